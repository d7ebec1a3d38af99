#!/usr/bin/env python
"""A few upscale_batch calls on pageable caller buffers (Lanczos-3, 1080p -> 4K), for a rocprofv3 --memory-copy-trace timeline:
   cd /tmp && rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d out -o t -- python3 tools/host_batch_trace.py [pinned]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

pinned = len(sys.argv) > 1 and sys.argv[1] == "pinned"
w, h = 1920, 1080
frames = [syn.gradient_frame(w, h, k).tobytes() for k in range(12)]
u = nsc.PyWgpuUpscaler("quality", "lanczos3")
u.initialize(w, h, 2 * w, 2 * h)
if pinned:
    src = [bytearray(f) for f in frames]
    dst = [bytearray(u.output_size) for _ in frames]
    pins = [nsc.PinnedBuffer(b) for b in src + dst]  # kept alive to the end
else:
    src = frames
    dst = [bytearray(u.output_size) for _ in frames]
u.upscale_batch_into(src, dst)
for rep in range(3):
    t0 = time.perf_counter()
    u.upscale_batch_into(src, dst)
    print(f"batch {rep}: {(time.perf_counter() - t0) / 12 * 1e3:.3f} ms/frame", flush=True)
    time.sleep(0.02)
