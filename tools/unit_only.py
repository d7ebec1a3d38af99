#!/usr/bin/env python
"""Launch only the one-launch pipeline step (unit kernel + its two edge-column passes) a few times, for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pattern = sys.argv[3] if len(sys.argv) > 3 else "gradient"
dev = torch.device("cuda:0")
w, h = 1920, 1080
frames = (syn.noise_stream_torch if pattern == "noise" else syn.gradient_stream_torch)(n + 1, w, h, dev)
pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
pipe.upscaler.set_option("rows_per_wave", 108)  # what the bench's 300-unit batches select (HipUpscaler::lanczos_x2_rows_per_wave)
mid, up_real, up_mid = pipe.alloc(n, dev)
for _ in range(reps):
    pipe.step_unit(frames, mid, up_real, up_mid, 0)
torch.cuda.synchronize()
print("ok", pipe.upscaler.kernel_variant)
