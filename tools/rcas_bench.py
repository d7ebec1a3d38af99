#!/usr/bin/env python
"""Time the RCAS pass alone on 4K frames (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
w, h = 3840, 2160
g = torch.Generator(device=dev)
g.manual_seed(1)
frames = torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(frames)
u = nsc.PyWgpuUpscaler("quality", "rcas")
u.initialize(w, h, w, h)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
torch.cuda.synchronize()
u.set_profiling(True)
for _ in range(5):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
nl, ms = u.profile_collect()
print(f"rcas 4K -> 4K, {n} frames per launch: {ms / nl / n * 1e3:.2f} us per frame ({u.kernel_variant})")
