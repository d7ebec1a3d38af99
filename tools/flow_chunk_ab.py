#!/usr/bin/env python
"""The motion step's default configuration with larger chunks (dev tool): FramePipeline.step_motion(chunk=c) with the estimator's internal
chunk limit lifted to match (NUS_FLOW_MAX_CHUNK_PAIRS, NUS_FLOW_WORKSPACE_GB): 100 (the product), 150, 300 units per chunk, interleaved."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import stream as S

os.environ["NUS_FLOW_WORKSPACE_GB"] = "24"
w, h, n = 1920, 1080, 300
dev = torch.device("cuda:0")
frames = S.SyntheticSource("gradient")(0, n + 1, w, h, dev)
pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
pipe.interp.set_mode("fma")
mid, up_real, up_mid = pipe.alloc(n, dev)
s = torch.cuda.current_stream().cuda_stream
for rnd in range(3):
    for c in (100, 150, 300):
        os.environ["NUS_FLOW_MAX_CHUNK_PAIRS"] = str(c)
        kw = dict(flow_mode="fast", pipelined=True, fused_warp=True, chunk=c)
        for _ in range(2):
            pipe.step_motion(frames, None, mid, up_real, up_mid, s, **kw)
        torch.cuda.synchronize()
        got = []
        for _ in range(5):
            t0 = time.perf_counter()
            pipe.step_motion(frames, None, mid, up_real, up_mid, s, **kw)
            torch.cuda.synchronize()
            got.append((time.perf_counter() - t0) * 1e3)
        got.sort()
        print(f"motion step, 300 units, default configuration, chunks of {c:3d} units (estimator chunks the same): {got[2]:7.2f} ms (min {got[0]:.2f})", flush=True)
