#!/usr/bin/env python
"""Timing of the fixed-factor kernels against rows per wave (dev tool): python tools/xs_probe.py [frames] [rows_per_wave ...]
NUS_DIMS=iwxih:owxoh picks another size (default 960x540:3840x2160), NUS_PATTERN=gradient the opaque stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ths = [int(v) for v in sys.argv[2:]] or [0]
dev = torch.device("cuda:0")
(iw, ih), (ow, oh) = [tuple(int(v) for v in part.split("x")) for part in os.environ.get("NUS_DIMS", "960x540:3840x2160").split(":")]
frames = (syn.gradient_stream_torch if os.environ.get("NUS_PATTERN") == "gradient" else syn.noise_stream_torch)(n, iw, ih, dev)
out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
for th in ths:
    u = nsc.PyWgpuUpscaler("quality", "lanczos3")
    if th:
        u.set_option("rows_per_wave", th)
    u.initialize(iw, ih, ow, oh)
    u.set_profiling(True)
    for _ in range(2):
        u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
    torch.cuda.synchronize()
    u.profile_collect()
    for _ in range(5):
        u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
    torch.cuda.synchronize()
    nl, ms = u.profile_collect()
    print(f"{u.kernel_variant} frames={n} rows_per_wave={th}: main kernel {ms/nl/n*1e3:.2f} us/frame")
