#!/usr/bin/env python
"""What in a long-lived bench process slows concurrent H2D + D2H copies and upscale_batch (dev tool): the same two measurements
after each thing bench.py does before its host-path leg."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

dev = torch.device("cuda:0")
w, h = 1920, 1080
frames = [syn.gradient_frame(w, h, k).tobytes() for k in range(12)]
u = nsc.PyWgpuUpscaler("quality", "lanczos3")
u.initialize(w, h, 2 * w, 2 * h)
outs = [bytearray(u.output_size) for _ in range(12)]


def measure(tag):
    c = bench.pcie_ceiling(torch, dev, u.input_size, u.output_size)
    u.upscale_batch_into(frames, outs)
    t0 = time.perf_counter()
    for _ in range(3):
        u.upscale_batch_into(frames, outs)
    b = (time.perf_counter() - t0) / 36 * 1e3
    t0 = time.perf_counter()
    for i in range(12):
        u.upscale_into(frames[i], outs[0])
    s = (time.perf_counter() - t0) / 12 * 1e3
    print(f"{tag:46s} both-directions {c['both_directions_ms_per_frame_pair']:.3f} ms  d2h {c['d2h_4k_frame_ms']:.3f} ms  "
          f"batch {b:.3f} ms/frame  single {s:.3f} ms", flush=True)


measure("fresh process")
import oracle
oracle.build()
a = oracle.gen_gradient(w, h, 0)
oracle.lanczos3(a, 2 * w, 2 * h, threads=0)
measure("after an all-cores OpenMP region")
time.sleep(1.0)
measure("... one second later")
pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
n = 300
fr = syn.gradient_stream_torch(n + 1, w, h, dev)
mid, up_real, up_mid = pipe.alloc(n, dev)
for _ in range(100):
    pipe.step_unit(fr, mid, up_real, up_mid, 0)
torch.cuda.synchronize()
measure("after 100 unit steps (25 GB resident)")
del fr, mid, up_real, up_mid
torch.cuda.empty_cache()
measure("after freeing the device buffers")
subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True)
measure("after a rocm-smi child")
flows = torch.empty((8, h, w, 2), dtype=torch.float32, device=dev)
fr = syn.gradient_stream_torch(9, w, h, dev)
mid, up_real, up_mid = pipe.alloc(8, dev)
pipe.step_motion(fr, flows, mid, up_real, up_mid, 0)
torch.cuda.synchronize()
measure("after a motion step (flow estimator stream)")
