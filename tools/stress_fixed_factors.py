#!/usr/bin/env python
"""Random sizes / filters / rows per wave through the fixed-factor resize kernels (x2, x3, x4, x3/2, x4/3; x5/4, x6/5, x5/3, x5/2, x7/2) against the oracle:
EXACT mode must be bit-exact, FMA mode within 1 LSB (dev tool, run on the GPU box: python tools/stress_fixed_factors.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nu_scaler_amd as nsc
import oracle as orc
rng = np.random.default_rng(7)
bad = 0
cases = 0
PQ = {"pq54": (5, 4), "pq65": (6, 5), "pq53": (5, 3), "pq52": (5, 2), "pq72": (7, 2)}
for kind in ("r32", "r43", "x3", "x4", "x2", "pq54", "pq65", "pq53", "pq52", "pq72"):
    for _ in range(14):
        if kind in PQ:  # groups of Q columns / rows, any number of them (strip joints, the direct and the turned stores)
            P, Q = PQ[kind]
            gw = int(rng.integers(-(-32 // Q), 1300 // Q)); gh = int(rng.integers(-(-12 // Q), 90 // Q + 2))
            w, h, ow, oh = Q * gw, Q * gh, P * gw, P * gh
        elif kind == "r32":
            w = 8 * int(rng.integers(4, 140)); h = 2 * int(rng.integers(8, 60)); ow, oh = 3 * w // 2, 3 * h // 2
        elif kind == "r43":
            w = 12 * int(rng.integers(4, 100)); h = 3 * int(rng.integers(6, 40)); ow, oh = 4 * w // 3, 4 * h // 3
        elif kind == "x3":
            w = 4 * int(rng.integers(4, 200)); h = int(rng.integers(16, 90)); ow, oh = 3 * w, 3 * h
        elif kind == "x4":
            w = 4 * int(rng.integers(4, 200)); h = int(rng.integers(16, 90)); ow, oh = 4 * w, 4 * h
        else:
            w = 4 * int(rng.integers(4, 300)); h = int(rng.integers(16, 120)); ow, oh = 2 * w, 2 * h
        alg, filt = [("lanczos3", 0), ("bicubic", 1), ("triangle", 2)][int(rng.integers(0, 3))]
        img = orc.gen_noise(w, h, int(rng.integers(1, 1000)))
        if rng.integers(0, 2):
            img[..., 3] = 255
        want = orc.resize(img, ow, oh, filt)
        u = nsc.PyWgpuUpscaler("quality", alg, lanczos_mode="exact")
        th = int(rng.integers(0, 50))
        if th:
            u.set_option("rows_per_wave", th)
        u.initialize(w, h, ow, oh)
        got = np.frombuffer(u.upscale(img.tobytes()), np.uint8).reshape(oh, ow, 4)
        uf = nsc.PyWgpuUpscaler("quality", alg); uf.initialize(w, h, ow, oh)
        gf = np.frombuffer(uf.upscale(img.tobytes()), np.uint8).reshape(oh, ow, 4)
        ok = np.array_equal(got, want) and np.abs(gf.astype(int) - want.astype(int)).max() <= 1
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", kind, alg, w, h, th, u.kernel_variant)
# nearest / bilinear (CPU form) on the fixed-ratio kernels: bit-exact
for (P, Q) in ((3, 2), (4, 3), (3, 1), (4, 1), (2, 1), (5, 4), (6, 5), (5, 3), (5, 2), (7, 2)):
    for _ in range(10):
        gw = int(rng.integers(1, 400)); gh = int(rng.integers(1, 120))
        w, h, ow, oh = Q * gw, Q * gh, P * gw, P * gh
        img = orc.gen_noise(w, h, int(rng.integers(1, 1000)))
        for alg, want in (("nearest", orc.nearest(img, ow, oh)), ("bilinear", orc.bilinear(img, ow, oh))):
            u = nsc.PyWgpuUpscaler("quality", alg); u.initialize(w, h, ow, oh)
            got = np.frombuffer(u.upscale(img.tobytes()), np.uint8).reshape(oh, ow, 4)
            cases += 1
            if not np.array_equal(got, want):
                bad += 1
                print("MISMATCH", alg, P, Q, w, h, u.kernel_variant)
print(f"{cases} cases, {bad} mismatches")
