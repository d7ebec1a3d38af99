#!/usr/bin/env python
"""Lanczos x2 kernel only, gradient stream, with an input format (for --pmc passes): lanczos_fmt.py <fmt> [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn
fmt = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
frames = syn.gradient_stream_torch(n, 1920, 1080, dev)
out = torch.empty((n, 2160, 3840, 4), dtype=torch.uint8, device=dev)
u = nsc.PyWgpuUpscaler("quality", "lanczos3"); u.set_input_format(fmt); u.initialize(1920, 1080, 3840, 2160)
for _ in range(3):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, 0)
torch.cuda.synchronize()
print("ok", u.kernel_variant)
