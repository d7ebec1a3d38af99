#!/usr/bin/env python
"""Large device-resident batches through the row-walking resize kernels (LDS-DMA row rings), every frame compared bit for bit with the
same frame upscaled alone and with a second run of the batch: a wave reading a ring slot that a later request has already
overwritten shows up as a few wrong pixels in a big batch and never on one frame (dev tool, run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
cases = [  # (iw, ih, ow, oh, frames)
    (2560, 1440, 3840, 2160, 24),   # x3/2
    (1280, 720, 1920, 1080, 96),    # x3/2
    (1920, 1080, 2560, 1440, 48),   # x4/3
    (1280, 720, 3840, 2160, 32),    # x3
    (960, 540, 3840, 2160, 32),     # x4
    (1600, 900, 1920, 1080, 64),    # general, 2 columns per lane (ring)
    (1536, 864, 3840, 2160, 24),    # general x2.5
    (1920, 1080, 3200, 1800, 24),   # general, union weights in LDS (no ring)
    (3840, 2160, 1920, 1080, 16),   # down-scaling, 58-column segments
    (2560, 1440, 1920, 1080, 32),   # down-scaling
    (1920, 1080, 3840, 2160, 48),   # x2
]
bad = 0
for iw, ih, ow, oh, n in cases:
    for mode in ("fma", "exact"):
        frames = syn.noise_stream_torch(n, iw, ih, dev)
        u = nsc.PyWgpuUpscaler("quality", "lanczos3", lanczos_mode=mode)
        u.initialize(iw, ih, ow, oh)
        a = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        one = torch.empty((1, oh, ow, 4), dtype=torch.uint8, device=dev)
        mism = 0
        for rep in range(3):
            u.upscale_device(frames.data_ptr(), a.data_ptr(), n, s)
            u.upscale_device(frames.data_ptr(), b.data_ptr(), n, s)
            torch.cuda.synchronize()
            mism += int((a != b).sum().item())
        for k in range(0, n, max(1, n // 8)):
            u.upscale_device(frames[k].data_ptr(), one.data_ptr(), 1, s)
            torch.cuda.synchronize()
            mism += int((one[0] != a[k]).sum().item())
        bad += mism != 0
        print(f"{iw}x{ih}->{ow}x{oh} {n:3d} frames {mode:5s} {u.kernel_variant:22s} differing bytes: {mism}", flush=True)
        del frames, a, b, one
        torch.cuda.empty_cache()
print(f"{2 * len(cases)} runs, {bad} with differences")
sys.exit(1 if bad else 0)
