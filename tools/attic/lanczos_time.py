#!/usr/bin/env python
"""Time the Lanczos x2 main kernel of the library named by NUS_LIB_PATH (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tag = sys.argv[2] if len(sys.argv) > 2 else ""
names = {"0": "full", "1": "no stores", "2": "no vertical MACs", "3": "no horizontal MACs", "4": "no lane exchange",
         "5": "no cvt/pack", "6": "stores only"}
dev = torch.device("cuda:0")
frames = syn.noise_stream_torch(n, 1920, 1080, dev)
out = torch.empty((n, 2160, 3840, 4), dtype=torch.uint8, device=dev)
u = nsc.PyWgpuUpscaler("quality", "lanczos3")
th = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if th:
    u.set_option("rows_per_wave", th)
u.initialize(1920, 1080, 3840, 2160)
for _ in range(2):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, 0)
torch.cuda.synchronize()
u.set_profiling(True)
for _ in range(5):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, 0)
nl, ms = u.profile_collect()
print(f"ablate {tag} ({names.get(tag, '?'):20s}) th={th:3d}: {ms/nl/n*1e3:7.2f} us/frame main kernel")
