#!/usr/bin/env python
"""Random sizes through the table-driven bilinear kernel (staged source rows on up-scales and on down-scaling by ~2; gathers elsewhere),
CPU form and WGSL form, RGBA and BGRA input, single frames and a device batch, against the oracle: bit-exact (dev tool, GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import nu_scaler_amd as nsc
import oracle as orc
from nu_scaler_amd.transfer import to_numpy as fetch  # noqa: E402  (device -> host through nus_download)

rng = np.random.default_rng(11)
bad = cases = 0
for k in range(80):
    w = int(rng.integers(4, 700)); h = int(rng.integers(1, 90))
    kind = k % 4
    if kind == 0:    # up-scale, any factor
        ow = 4 * int(rng.integers((w + 3) // 4, (3 * w) // 4 + 2)); oh = int(rng.integers(h, 3 * h + 2))
    elif kind == 1:  # around /2
        ow = 4 * max(1, int(round(w / 2 / 4 * rng.uniform(0.95, 1.08)))); oh = max(1, int(round(h / 2 * rng.uniform(0.9, 1.1))))
    elif kind == 2:  # other down-scales
        ow = 4 * max(1, int(w / 4 / rng.uniform(1.1, 3.5))); oh = max(1, int(h / rng.uniform(1.0, 3.0)))
    else:            # x1 .. x1.1 (the staged row's widest reach)
        ow = 4 * ((w + 3) // 4 + int(rng.integers(0, 3))); oh = h + int(rng.integers(0, 3))
    if ow < 4:
        continue
    img = orc.gen_noise(w, h, int(rng.integers(1, 1000)))
    for variant, ref in (("cpu", orc.bilinear), ("wgsl", orc.bilinear_wgsl)):
        u = nsc.PyWgpuUpscaler("quality", "bilinear", bilinear_variant=variant)
        u.set_option("force_general", 1)
        u.initialize(w, h, ow, oh)
        want = ref(img, ow, oh)
        got = np.frombuffer(u.upscale(img.tobytes()), np.uint8).reshape(oh, ow, 4)
        cases += 1
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH", variant, w, h, ow, oh, u.kernel_variant, int(np.abs(got.astype(int) - want.astype(int)).max()))
    if k % 5 == 0 and (w * h) % 4 == 0:  # a device batch of three frames equals the single frames (frame sizes must be multiples of 16 B)
        frames = np.stack([img, img[::-1].copy(), orc.gen_noise(w, h, 5)])
        d_in = torch.from_numpy(frames).cuda()
        d_out = torch.empty((3, oh, ow, 4), dtype=torch.uint8, device="cuda")
        u = nsc.PyWgpuUpscaler("quality", "bilinear")
        u.set_option("force_general", 1)
        u.initialize(w, h, ow, oh)
        u.upscale_device(d_in.data_ptr(), d_out.data_ptr(), 3)
        torch.cuda.synchronize()
        got = fetch(d_out)
        for j in range(3):
            cases += 1
            if not np.array_equal(got[j], orc.bilinear(frames[j], ow, oh)):
                bad += 1
                print("MISMATCH batch", j, w, h, ow, oh)
print(f"{cases} cases, {bad} mismatches")
