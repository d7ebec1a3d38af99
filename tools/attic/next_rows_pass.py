#!/usr/bin/env python
"""One pass over the 'next row' kernels (FSR1 pair, x4 / any-factor resize, flow front end, warp with flow, swizzle)
for a rocprofv3 --kernel-trace --stats summary."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
n = 16
f1080 = syn.gradient_stream_torch(n + 1, 1920, 1080, dev)
out4k = torch.empty((n, 2160, 3840, 4), dtype=torch.uint8, device=dev)
for alg in ("fsr1", "easu", "bicubic"):
    u = nsc.PyWgpuUpscaler("quality", alg); u.initialize(1920, 1080, 3840, 2160)
    for _ in range(3):
        u.upscale_device(f1080.data_ptr(), out4k.data_ptr(), n, s)
for (iw, ih) in ((960, 540), (1280, 720), (2560, 1440)):
    fin = syn.gradient_stream_torch(n, iw, ih, dev)
    u = nsc.PyWgpuUpscaler("quality", "lanczos3"); u.initialize(iw, ih, 3840, 2160)
    for _ in range(3):
        u.upscale_device(fin.data_ptr(), out4k.data_ptr(), n, s)
fe = nsc.FlowEstimator()
flow = torch.empty((1080, 1920, 2), dtype=torch.float32, device=dev)
fb = 1920 * 1080 * 4
for k in range(4):
    fe.estimate_device(f1080.data_ptr() + k * fb, f1080.data_ptr() + (k + 1) * fb, 1920, 1080, flow.data_ptr(), s)
it = nsc.WgpuFrameInterpolator()
mid = torch.empty((1080, 1920, 4), dtype=torch.uint8, device=dev)
for _ in range(3):
    it.interpolate_device(f1080.data_ptr(), fb, f1080.data_ptr() + fb, fb, flow.data_ptr(), 1920, 1080, 0.5, mid.data_ptr(), 1, s)
nsc.swizzle_bgra_to_rgba_device(f1080.data_ptr(), f1080.data_ptr(), 1920 * 1080, s)
torch.cuda.synchronize()
print("ok")
