#!/usr/bin/env python
"""Where does the FAST flow differ from the oracle's? (dev tool: flow_fast_debug.py w h levels coarse refine)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import nu_scaler_amd as nsc

from nu_scaler_amd import hostmem  # noqa: E402

hostmem.route_tensor_cpu_through_pinned_staging()  # device -> pinned staging -> numpy (nu_scaler_amd/hostmem.py)
import oracle
from test_flow import _smooth
w, h, levels, coarse, refine = [int(v) for v in sys.argv[1:6]]
frames = np.stack([_smooth(w, h, 1.3 * k) for k in range(4)])
fe = nsc.FlowEstimator(levels=levels, coarse_iterations=coarse, refine_iterations=refine)
fe.set_mode("fast")
dev = torch.device("cuda:0")
d = torch.from_numpy(frames).to(dev)
fl = torch.empty((3, h, w, 2), dtype=torch.float32, device=dev)
fe.estimate_device_stream(d.data_ptr(), 4, w, h, fl.data_ptr(), 0)
torch.cuda.synchronize()
got = fl.cpu().numpy()
for k in range(3):
    want = oracle.flow_estimate(frames[k], frames[k + 1], levels, coarse, refine, fe.lambda_)
    e = np.abs(got[k] - want).max(-1)
    y, x = np.unravel_index(e.argmax(), e.shape)
    print(f"pair {k}: max {e.max():.3e} at (y={y}, x={x}); rows with err>1e-3: {np.flatnonzero((e > 1e-3).any(1))[:12]} cols: {np.flatnonzero((e > 1e-3).any(0))[:12]}  count {(e > 1e-3).sum()}")
