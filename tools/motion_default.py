#!/usr/bin/env python
"""FramePipeline.step_motion in its default configuration (FAST flow, two-stream pipeline over 100-unit chunks, one entry point,
Rg16Float hand-off) on a device-resident 1080p stream, 300 units: ms per step, median of 5 (dev tool; NUS_LIB_PATH picks an A/B library)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import stream as S

w, h, n = 1920, 1080, 300
dev = torch.device("cuda:0")
frames = S.SyntheticSource("gradient")(0, n + 1, w, h, dev)
pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
pipe.interp.set_mode("fma")
mid, up_real, up_mid = pipe.alloc(n, dev)
s = torch.cuda.current_stream().cuda_stream
kw = dict(flow_mode="fast", pipelined=True, fused_warp=True)
for fmt in (None, "f32"):
    for _ in range(2):
        pipe.step_motion(frames, None, mid, up_real, up_mid, s, flow_format=fmt, **kw)
    torch.cuda.synchronize()
    got = []
    for _ in range(5):
        t0 = time.perf_counter()
        pipe.step_motion(frames, None, mid, up_real, up_mid, s, flow_format=fmt, **kw)
        torch.cuda.synchronize()
        got.append((time.perf_counter() - t0) * 1e3)
    got.sort()
    print(f"motion step, 300 units, default configuration, hand-off {'Rg16Float' if fmt is None else 'f32'}: {got[2]:7.2f} ms  (min {got[0]:.2f})", flush=True)
