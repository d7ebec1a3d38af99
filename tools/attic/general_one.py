#!/usr/bin/env python
"""Launch one algorithm at one size a few times (for rocprofv3 --pmc passes): general_one.py iw ih ow oh alg [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn
iw, ih, ow, oh = (int(v) for v in sys.argv[1:5]); alg = sys.argv[5]; n = int(sys.argv[6]) if len(sys.argv) > 6 else 64
dev = torch.device("cuda:0")
frames = syn.noise_stream_torch(n, iw, ih, dev)
out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
u = nsc.PyWgpuUpscaler("quality", alg); u.initialize(iw, ih, ow, oh)
for _ in range(3):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, 0)
torch.cuda.synchronize()
print("ok", u.kernel_variant)
