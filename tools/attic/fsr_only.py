#!/usr/bin/env python
"""One FSR1 variant alone, a few launches (for counter passes: tools/sq_kernel.sh k_fsr1 -- tools/fsr_only.py easu 1 noise 64 2).
usage: fsr_only.py easu|fsr1|rcas fast(0|1) pattern frames launches"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

alg, fast, pattern, n, reps = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda:0")
iw, ih, ow, oh = (3840, 2160, 3840, 2160) if alg == "rcas" else (1920, 1080, 3840, 2160)
frames = (syn.gradient_stream_torch if pattern == "gradient" else syn.noise_stream_torch)(n, iw, ih, dev)
out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
u = nsc.PyWgpuUpscaler("quality", alg)
if fast:
    u.set_option("fsr_fast", 1)
u.initialize(iw, ih, ow, oh)
s = torch.cuda.current_stream().cuda_stream
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
torch.cuda.synchronize()
a.record()
for _ in range(reps):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
b.record()
torch.cuda.synchronize()
print(f"{alg} fast={fast} {pattern} {u.kernel_variant}: {a.elapsed_time(b) / reps / n * 1e3:.2f} us/frame")
