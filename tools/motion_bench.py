#!/usr/bin/env python
"""Time FramePipeline.step_motion on a device-resident 1080p stream (dev tool): stage after stage against the two-stream
pipeline over chunks.  motion_bench.py [units] [chunk ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import stream as S

w, h = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
chunks = [int(c) for c in sys.argv[2:]] or [100]
dev = torch.device("cuda:0")
frames = S.SyntheticSource("gradient")(0, n + 1, w, h, dev)
pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
pipe.interp.set_mode("fma")
mid, up_real, up_mid = pipe.alloc(n, dev)
flows = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream


def timed(**kw):
    fl = None if kw.get("fused_warp") else flows  # fused: the flow never goes to HBM
    return _timed(fl, **kw)


def _timed(flows, **kw):
    pipe.step_motion(frames, flows, mid, up_real, up_mid, s, flow_mode="fast", **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        pipe.step_motion(frames, flows, mid, up_real, up_mid, s, flow_mode="fast", **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3


for rnd in range(2):
    print(f"motion step, {n} units, stage after stage: {timed():7.2f} ms", flush=True)
    for c in chunks:
        pipe._motion_real_first = False
        print(f"motion step, {n} units, pipelined, chunks of {c:3d}, real frames per chunk: {timed(pipelined=True, chunk=c):7.2f} ms", flush=True)
        pipe._motion_real_first = True
        print(f"motion step, {n} units, pipelined, chunks of {c:3d}: {timed(pipelined=True, chunk=c):7.2f} ms", flush=True)
    for env in ("0", "1"):  # one entry point (estimator, then the warp kernel behind it) / the warp inside the last Jacobi launch
        os.environ["NUS_HS_FUSED_WARP"] = env
        what = "warp inside the last Jacobi launch" if env == "1" else "one call, warp kernel behind the estimator"
        print(f"motion step, {n} units, {what}, stage after stage: {timed(fused_warp=True, flow_format='f32'):7.2f} ms", flush=True)
        for c in chunks:
            print(f"motion step, {n} units, {what}, pipelined, chunks of {c:3d}: {timed(fused_warp=True, pipelined=True, chunk=c, flow_format='f32'):7.2f} ms", flush=True)
    os.environ.pop("NUS_HS_FUSED_WARP", None)
    for c in chunks:
        print(f"motion step, {n} units, one call, Rg16Float between estimator and warp, pipelined, chunks of {c:3d}: "
              f"{timed(fused_warp=True, pipelined=True, chunk=c, flow_format='f16'):7.2f} ms", flush=True)
