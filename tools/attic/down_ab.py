#!/usr/bin/env python
"""Down-scaling kernel: us per frame by output columns per wave (option down_seg_width; 0 = the host's choice), Lanczos-3,
gradient and noise input.  usage: down_ab.py iw ih ow oh frames width[,width...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

iw, ih, ow, oh, n = (int(v) for v in sys.argv[1:6])
widths = [int(v) for v in sys.argv[6].split(",")]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
for pat, gen in (("gradient", syn.gradient_stream_torch), ("noise", syn.noise_stream_torch)):
    frames = gen(n, iw, ih, dev)
    res = []
    for rep in range(2):
        for sw in widths:
            u = nsc.PyWgpuUpscaler("quality", "lanczos3")
            u.set_option("down_seg_width", sw)
            u.initialize(iw, ih, ow, oh)
            for _ in range(3 if rep else 8):
                u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 5 / n * 1e3)
            if rep:
                res.append(f"w={sw}: {best:6.2f}")
    print(f"{iw}x{ih}->{ow}x{oh} {pat:8s} {u.kernel_variant}  " + "   ".join(res), flush=True)
