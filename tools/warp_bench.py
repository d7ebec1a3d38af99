#!/usr/bin/env python
"""Time k_warp_blend_flow on a device-resident 1080p stream: constant flow and a smooth random flow (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

w, h, n = 1920, 1080, 100
dev = torch.device("cuda:0")
frames = syn.gradient_stream_torch(n + 1, w, h, dev)
mid = torch.empty((n, h, w, 4), dtype=torch.uint8, device=dev)
fb = w * h * 4
it = nsc.WgpuFrameInterpolator()
s = torch.cuda.current_stream().cuda_stream
const = torch.zeros((n, h, w, 2), dtype=torch.float32, device=dev)
const[..., 0] = -1.0
g = torch.Generator(device=dev).manual_seed(3)
coarse = torch.randn((n, 2, h // 40, w // 40), generator=g, device=dev) * 4.0
smooth = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear").permute(0, 2, 3, 1).contiguous()
cases = [("constant (-1, 0)", const, "f32"), ("smooth random, sigma 4 px", smooth, "f32"),
         ("smooth random, f16 flow field", smooth.to(torch.float16), "f16")]
for name, flow, fmt, mode in [c + (m,) for c in cases for m in ("exact", "fma")]:
    it.set_flow_format(fmt)
    it.set_mode(mode)
    run = lambda: it.interpolate_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, flow.data_ptr(), w, h, 0.5, mid.data_ptr(), n, s)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 / n * 1e3
    nb = (5 if fmt == "f32" else 4) * fb
    print(f"warp_blend_flow {mode:5s} {name:32s} {us:7.2f} us/pair  {nb / us / 1e6:5.2f} TB/s algorithmic = {nb / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s")
