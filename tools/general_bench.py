#!/usr/bin/env python
"""Time the any-scale (table-driven) kernels at a non-x2 factor, e.g. 720p -> 1080p (x1.5) (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

iw, ih, ow, oh = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (1280, 720, 1920, 1080)))
n = int(sys.argv[5]) if len(sys.argv) > 5 else 64
dev = torch.device("cuda:0")
frames = (syn.gradient_stream_torch if os.environ.get("NUS_PATTERN") == "gradient" else syn.noise_stream_torch)(n, iw, ih, dev)
out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
alg_bytes = (iw * ih + ow * oh) * 4
for alg in ("nearest", "bilinear", "lanczos3", "bicubic"):
    u = nsc.PyWgpuUpscaler("quality", alg)
    if os.environ.get("NUS_GENERAL"):
        u.set_option("force_general", 1)  # the table-driven kernels even where a fixed-factor kernel exists
    u.initialize(iw, ih, ow, oh)
    for _ in range(2):
        u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 5 / n * 1e3
    print(f"{iw}x{ih}->{ow}x{oh} {alg:9s} {u.kernel_variant:22s} {us:8.2f} us/frame  {alg_bytes/us/1e6:6.2f} TB/s algorithmic")
