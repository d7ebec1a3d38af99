#!/usr/bin/env python
"""x5/4, x6/5, x5/3, x5/2: the P/Q register-window kernel next to the any-scale kernel it replaces (dev tool).

usage: pq_bench.py [frames]      NUS_PQ_TH=<rows per wave> overrides the host's choice
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
th = int(os.environ.get("NUS_PQ_TH", "0"))
shapes = [(1536, 864, 1920, 1080), (1600, 900, 1920, 1080), (1920, 1080, 3200, 1800), (1536, 864, 3840, 2160), (1096, 616, 3836, 2156),
          (1920, 1080, 2688, 1512), (1920, 1080, 3072, 1728), (1920, 1080, 3456, 1944)]
for pattern in ("gradient", "noise"):
    for iw, ih, ow, oh in shapes:
        frames = (syn.gradient_stream_torch if pattern == "gradient" else syn.noise_stream_torch)(n, iw, ih, dev)
        out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
        alg_bytes = (iw * ih + ow * oh) * 4
        for alg in ("lanczos3", "bicubic"):
            for mode in ("fma", "exact"):
                line = f"{pattern:8s} {iw}x{ih}->{ow}x{oh} {alg:8s} {mode:5s}"
                # bicubic: the kernel's 4-tap form (round 6, the default), its 6-tap form (option pq_narrow 0), the any-scale kernel
                for opts in (({}, {"pq_narrow": 0}, {"force_general": 1}) if alg == "bicubic" else ({}, {"force_general": 1})):
                    u = nsc.PyWgpuUpscaler("quality", alg, lanczos_mode=mode)
                    for k, v in opts.items():
                        u.set_option(k, v)
                    if th and not opts:
                        u.set_option("rows_per_wave", th)
                    u.initialize(iw, ih, ow, oh)
                    for _ in range(2):
                        u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    best = 1e30
                    for _ in range(3):  # the best of three timed groups of five launches
                        a.record()
                        for _ in range(5):
                            u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
                        b.record()
                        torch.cuda.synchronize()
                        best = min(best, a.elapsed_time(b) / 5 / n * 1e3)
                    us = best
                    tag = u.kernel_variant + ("/4tap" if u.get_option("pq_narrow_active") else "")
                    line += f"  {tag:24s} {us:7.2f} us/frame {alg_bytes / us / 1e6:5.2f} TB/s"
                print(line, flush=True)
