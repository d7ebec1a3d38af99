#!/usr/bin/env python
"""Dev switch NUS_HS_L0_HALF_BETWEEN=1 (the flow BETWEEN the finest level's two Jacobi launches as Rg16Float): what it does to the FAST
contract at 1080p (flow against the oracle's, the interpolated frame against the oracle's) and to the motion step's time.  One process per
setting (the switch is read per launch, but the comparison wants identical conditions).  l0_half_between.py accuracy|time"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd.transfer import to_device as put, to_numpy as fetch

what = sys.argv[1] if len(sys.argv) > 1 else "accuracy"
w, h = 1920, 1080
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
if what == "accuracy":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle
    from test_flow import _smooth

    oracle.build()
    n = 33  # enough pairs for the FAST streamed kernels (a single pair runs the exact LDS-tile kernels)
    rng = np.random.default_rng(7)
    imgs = []
    for k in range(n):
        img = _smooth(w, h, 1.75 * k)
        img[..., :3] = np.clip(img[..., :3].astype(np.int16) + rng.integers(-3, 4, size=(h, w, 3)), 0, 255).astype(np.uint8)
        imgs.append(img)
    fe = nsc.FlowEstimator(levels=3, coarse_iterations=50, refine_iterations=10)
    picks = (0, 17, n - 2)
    want = {k: oracle.flow_estimate(imgs[k], imgs[k + 1], 3, 50, 10, fe.lambda_) for k in picks}
    ref = {k: oracle.warp_blend(imgs[k], imgs[k + 1], want[k], 0.5) for k in picks}
    fe.set_mode("fast")
    frames = put(np.stack(imgs))
    for sw in ("0", "1"):
        os.environ["NUS_HS_L0_HALF_BETWEEN"] = sw
        flow = torch.empty((n - 1, h, w, 2), dtype=torch.float32, device=dev)
        mid = torch.empty((n - 1, h, w, 4), dtype=torch.uint8, device=dev)
        fe.interpolate_device_stream(frames.data_ptr(), n, w, h, 0.5, mid.data_ptr(), flow.data_ptr(), s, "f32")
        torch.cuda.synchronize()
        for k in picks:
            err = np.abs(fetch(flow[k]) - want[k])
            d = np.abs(fetch(mid[k]).astype(np.int16) - ref[k].astype(np.int16))
            print(f"half between the level-0 launches = {sw}, pair {k:2d}: flow max |error| {err.max():.2e} px, mean {err.mean():.2e}; interpolated frame "
                  f"max |d| {int(d.max())}, differing {100 * float((d > 0).mean()):.4f} %", flush=True)
else:
    import time

    from nu_scaler_amd import stream as S

    n = 300
    frames = S.SyntheticSource("gradient")(0, n + 1, w, h, dev)
    pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
    pipe.interp.set_mode("fma")
    mid, up_real, up_mid = pipe.alloc(n, dev)
    kw = dict(flow_mode="fast", pipelined=True, fused_warp=True)
    for rnd in range(3):
        for sw in ("0", "1"):
            os.environ["NUS_HS_L0_HALF_BETWEEN"] = sw
            for _ in range(2):
                pipe.step_motion(frames, None, mid, up_real, up_mid, s, **kw)
            torch.cuda.synchronize()
            got = []
            for _ in range(5):
                t0 = time.perf_counter()
                pipe.step_motion(frames, None, mid, up_real, up_mid, s, **kw)
                torch.cuda.synchronize()
                got.append((time.perf_counter() - t0) * 1e3)
            got.sort()
            print(f"motion step, 300 units, default configuration, half between the level-0 launches = {sw}: {got[2]:7.2f} ms (min {got[0]:.2f})", flush=True)
