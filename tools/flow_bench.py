#!/usr/bin/env python
"""Time the optical-flow front end at 1080p on device-resident frames (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

w, h = 1920, 1080
levels, ci, ri = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 3), (2, 50), (3, 10)))
dev = torch.device("cuda:0")
frames = syn.gradient_stream_torch(2, w, h, dev) // 2 + syn.noise_stream_torch(2, w, h, dev) // 2
flow = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
fe = nsc.FlowEstimator(levels=levels, coarse_iterations=ci, refine_iterations=ri)
s = torch.cuda.current_stream().cuda_stream
a, b = frames[0].data_ptr(), frames[1].data_ptr()
modes = [int(m) for m in sys.argv[4:]] or [1]  # nus_flow_set_tiled: 1 kernels by size, 2 LDS tiles, 3 streamed; 9 = FAST arithmetic
for mode in modes:
    fe.set_mode("fast" if mode == 9 else "exact")
    fe.set_tiled(1 if mode == 9 else mode)
    for _ in range(2):
        fe.estimate_device(a, b, w, h, flow.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        fe.estimate_device(a, b, w, h, flow.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    print(f"flow estimate 1080p levels={levels} coarse={ci} refine={ri} kernel mode {mode}: {e0.elapsed_time(e1)/n*1e3:.1f} us per pair")
