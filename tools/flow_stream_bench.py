#!/usr/bin/env python
"""Time nus_flow_estimate_device_stream on a device-resident 1080p stream (dev tool): flow_stream_bench.py [frames] [modes: 1 by size, 2 LDS tiles, 3 streamed, 9 = FAST arithmetic (nus_flow_set_mode), 19 = FAST with the shifting pass ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

w, h = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65
dev = torch.device("cuda:0")
frames = syn.gradient_stream_torch(n, w, h, dev) // 2 + syn.noise_stream_torch(n, w, h, dev) // 2
flows = torch.empty((n - 1, h, w, 2), dtype=torch.float32, device=dev)
fe = nsc.FlowEstimator(levels=3, coarse_iterations=50, refine_iterations=10)
s = torch.cuda.current_stream().cuda_stream
modes = [int(m) for m in sys.argv[2:]] or [1]
ref = None
for rnd in range(3 if len(modes) > 1 else 1):  # interleaved rounds when several kernels are compared
    for mode in modes:
        fe.set_mode("fast" if mode in (9, 19) else "exact")
        fe.set_tiled(1 if mode in (9, 19) else mode)
        if mode == 19:  # FAST with the shifting form of the pass (rounds 3-4) where the ring form is the product's
            os.environ["NUS_HS_FAST_SHIFT"] = "1"
        else:
            os.environ.pop("NUS_HS_FAST_SHIFT", None)
        fe.estimate_device_stream(frames.data_ptr(), n, w, h, flows.data_ptr(), s)
        torch.cuda.synchronize()
        if ref is None:
            ref = flows.clone()
        same = bool(torch.equal(flows, ref))
        maxdiff = float((flows - ref).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            fe.estimate_device_stream(frames.data_ptr(), n, w, h, flows.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        print(f"flow stream 1080p, {n} frames, kernel mode {mode}: {e0.elapsed_time(e1) / reps / (n - 1) * 1e3:.1f} us per pair"
              f"{'' if same else f'  output differs from the first mode: max |d| = {maxdiff:.2e} px'}", flush=True)
