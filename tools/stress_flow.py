#!/usr/bin/env python
"""Random sizes / pyramid depths / iteration counts / batch lengths through nus_flow_estimate_device_stream in every kernel mode
(by size, streamed, LDS tiles) against the oracle, bit for bit (dev tool, run on the GPU box: python tools/stress_flow.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nu_scaler_amd as nsc
from nu_scaler_amd.transfer import to_numpy as fetch  # noqa: E402  (device -> host through nus_download)
import oracle as orc
rng = np.random.default_rng(11)
bad = cases = 0
dev = torch.device("cuda:0")
for _ in range(24):
    w = int(rng.integers(20, 420)); h = int(rng.integers(12, 300)); levels = int(rng.integers(1, 5))
    ci = int(rng.integers(0, 24)); ri = int(rng.integers(0, 13)); n = int(rng.integers(2, 6))
    frames = np.stack([orc.gen_noise(w, h, int(rng.integers(1, 999))) for _ in range(n)])
    fe = nsc.FlowEstimator(levels=levels, coarse_iterations=ci, refine_iterations=ri)
    want = [orc.flow_estimate(frames[k], frames[k + 1], levels, ci, ri, fe.lambda_) for k in range(n - 1)]
    d_frames = torch.from_numpy(frames).to(dev)
    d_flows = torch.empty((n - 1, h, w, 2), dtype=torch.float32, device=dev)
    for mode in (1, 3, 2):
        fe.set_tiled(mode)
        d_flows.fill_(float("nan"))
        fe.estimate_device_stream(d_frames.data_ptr(), n, w, h, d_flows.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = fetch(d_flows)
        ok = all(np.array_equal(got[k], want[k]) for k in range(n - 1))
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", w, h, levels, ci, ri, n, mode)
print(f"{cases} cases, {bad} mismatches")

# round 5: FAST mode, the ring form of the pass against the shifting form (bit for bit) and against the oracle (1e-3 px), random shapes
for _ in range(16):
    w = int(rng.integers(20, 700)); h = int(rng.integers(12, 400)); levels = int(rng.integers(1, 4))
    ci = int(rng.integers(0, 40)); ri = int(rng.integers(0, 13)); n = int(rng.integers(2, 5))
    frames = np.stack([orc.gen_noise(w, h, int(rng.integers(1, 999))) for _ in range(n)])
    d_frames = torch.from_numpy(frames).to(dev)
    got = {}
    for form in ("ring", "shift"):
        if form == "shift":
            os.environ["NUS_HS_FAST_SHIFT"] = "1"
        else:
            os.environ.pop("NUS_HS_FAST_SHIFT", None)
        fe = nsc.FlowEstimator(levels=levels, coarse_iterations=ci, refine_iterations=ri)
        fe.set_mode("fast")
        fe.set_tiled(3)
        d_flows = torch.full((n - 1, h, w, 2), float("nan"), dtype=torch.float32, device=dev)
        fe.estimate_device_stream(d_frames.data_ptr(), n, w, h, d_flows.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got[form] = fetch(d_flows)
    os.environ.pop("NUS_HS_FAST_SHIFT", None)
    want = np.stack([orc.flow_estimate(frames[k], frames[k + 1], levels, ci, ri, fe.lambda_) for k in range(n - 1)])
    cases += 1
    err = float(np.abs(got["ring"].astype(np.float64) - want).max())
    if not np.array_equal(got["ring"], got["shift"]) or not err <= 1e-3:
        bad += 1
        print("FAST MISMATCH", w, h, levels, ci, ri, n, "ring == shift:", np.array_equal(got["ring"], got["shift"]), "max err", err)
print("fast-mode cases done; total", cases, "bad", bad)
