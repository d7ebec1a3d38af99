#!/usr/bin/env python
"""Launch only the Lanczos-3 x2 kernel a few times (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
pattern = sys.argv[3] if len(sys.argv) > 3 else "noise"
frames = (syn.noise_stream_torch if pattern == "noise" else syn.gradient_stream_torch)(n, 1920, 1080, dev)
out = torch.empty((n, 2160, 3840, 4), dtype=torch.uint8, device=dev)
u = nsc.PyWgpuUpscaler("quality", "lanczos3")
u.initialize(1920, 1080, 3840, 2160)
for _ in range(reps):
    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, 0)
torch.cuda.synchronize()
print("ok", u.kernel_variant)
