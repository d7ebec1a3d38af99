#!/usr/bin/env python
"""Randomised parity sweep of the one-launch pipeline step (dev tool): random sizes (ragged strips and row blocks), rows per
wave, batch lengths, t, wave order, Lanczos mode, filter, channel order and content (opaque / alpha / mixed); every output
buffer against the three separate stages bit for bit, and in EXACT mode against the oracle.  usage: stress_unit_step.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import nu_scaler_amd as nsc

from nu_scaler_amd.transfer import to_numpy as fetch  # noqa: E402  (device -> host through nus_download)

import oracle

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
bad = 0
for case in range(cases):
    w = int(rng.integers(4, 300)) * 4
    h = int(rng.integers(16, 120))
    n = int(rng.integers(1, 7))
    th = int(rng.choice([0, 6, 7, 12, 18, 36]))
    t = float(rng.choice([0.5, 0.5, 0.25, 0.3, 0.8]))
    order = int(rng.integers(0, 2))
    mode = str(rng.choice(["fma", "exact"]))
    alg = str(rng.choice(["lanczos3", "lanczos3", "bicubic", "triangle"]))
    fmt = str(rng.choice(["rgba", "rgba", "bgra", "rgbx"]))
    content = str(rng.choice(["noise", "opaque", "mixed", "flat", "flat"]))
    frames_np = np.stack([oracle.gen_noise(w, h, int(rng.integers(1, 1 << 30))) for _ in range(n + 1)])
    if content == "flat":  # round 5: regions of one alpha each (any value), the boundaries at random rows and columns, one frame out of step
        a0, a1, a2 = (int(v) for v in rng.integers(0, 256, 3))
        r1, r2 = sorted(int(v) for v in rng.integers(0, h, 2))
        c1 = int(rng.integers(0, w))
        frames_np[..., 3] = a0
        frames_np[:, r1:r2, :, 3] = a1
        frames_np[:, :, c1:, 3] = a2
        frames_np[int(rng.integers(0, n + 1)), :, :, 3] = int(rng.integers(0, 256))
    elif content != "noise":
        frames_np[..., 3] = 255
    if content == "mixed":
        k = int(rng.integers(0, n + 1))
        frames_np[k, h // 3:, :, 3] = 77
    frames = torch.from_numpy(frames_np).to(dev)
    fb = w * h * 4
    u = nsc.PyWgpuUpscaler("quality", alg, lanczos_mode=mode)
    if th:
        u.set_option("rows_per_wave", th)
    u.set_option("unit_order", order)
    u.set_input_format(fmt)
    u.initialize(w, h, 2 * w, 2 * h)
    it = nsc.WgpuFrameInterpolator()
    it.set_input_format(fmt)
    want_mid = torch.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
    it.interpolate_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, 0, w, h, t, want_mid.data_ptr(), n, s)
    want_real = torch.zeros((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
    u.upscale_device(frames.data_ptr(), want_real.data_ptr(), n, s)
    u.set_input_format("rgbx" if fmt == "rgbx" else "rgba")  # the in-between frames are RGBA (alpha already forced by an X format)
    want_up_mid = torch.zeros_like(want_real)
    u.upscale_device(want_mid.data_ptr(), want_up_mid.data_ptr(), n, s)
    u.set_input_format(fmt)
    mid = torch.zeros_like(want_mid)
    up_real, up_mid = torch.zeros_like(want_real), torch.zeros_like(want_real)
    u.upscale_unit_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, t, mid.data_ptr(), up_real.data_ptr(), up_mid.data_ptr(), n, s)
    torch.cuda.synchronize()
    ok = torch.equal(mid, want_mid) and torch.equal(up_real, want_real) and torch.equal(up_mid, want_up_mid)
    if ok and mode == "exact" and fmt == "rgba":
        filt = {"lanczos3": 0, "bicubic": 1, "triangle": 2}[alg]
        k = int(rng.integers(0, n))
        m = oracle.warp_blend(frames_np[k], frames_np[k + 1], None, t)
        ok = (np.array_equal(fetch(mid[k]), m) and np.array_equal(fetch(up_real[k]), oracle.resize(frames_np[k], 2 * w, 2 * h, filt))
              and np.array_equal(fetch(up_mid[k]), oracle.resize(m, 2 * w, 2 * h, filt)))
    if not ok:
        bad += 1
        print("MISMATCH", dict(w=w, h=h, n=n, th=th, t=t, order=order, mode=mode, alg=alg, fmt=fmt, content=content), flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
