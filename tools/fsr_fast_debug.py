#!/usr/bin/env python
"""Where does FAST EASU differ from the oracle by more than one count? (dev tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import nu_scaler_amd as nsc
import oracle

w, h, ow, oh = [int(v) for v in (sys.argv[1:5] or (100, 70, 257, 131))]
for name, img in (("noise", oracle.gen_noise(w, h, 77)), ("gradient", oracle.gen_gradient(w, h))):
    want = oracle.fsr_easu(img, ow, oh, 0.0)
    u = nsc.PyWgpuUpscaler("quality", "easu")
    u.set_option("fsr_fast", 1)
    u.set_sharpness(0.0, -1.0)
    u.initialize(w, h, ow, oh)
    got = np.frombuffer(u.upscale(img.tobytes()), np.uint8).reshape(oh, ow, 4)
    d = np.abs(got.astype(int) - want.astype(int))
    print(name, "max", d.max(), "count>1", int((d > 1).sum()), "count>0", int((d > 0).sum()), "of", d.size)
    ys, xs, cs = np.nonzero(d > 1)
    for y, x, c in list(zip(ys, xs, cs))[:8]:
        cx, cy = (x + 0.5) * np.float32(w) / np.float32(ow), (y + 0.5) * np.float32(h) / np.float32(oh)
        print(f"  out ({x},{y}) ch {c}: got {got[y, x, c]} want {want[y, x, c]}  src ({cx:.4f},{cy:.4f}) tile ({x // 64},{y // 32}) in-tile ({x % 64},{y % 32})")
