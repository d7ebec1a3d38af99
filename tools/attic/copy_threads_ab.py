#!/usr/bin/env python
"""Host path by copy-pool size (NUS_COPY_THREADS, read once per process): upscale(), upscale_batch, interpolate_py, 1080p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn
w, h = 1920, 1080
frames = [syn.gradient_frame(w, h, k).tobytes() for k in range(13)]
u = nsc.PyWgpuUpscaler("quality", "lanczos3")
u.initialize(w, h, 2 * w, 2 * h)
out = bytearray(u.output_size)
bufs = [bytearray(u.output_size) for _ in range(12)]
u.upscale_into(frames[0], out)
u.upscale_batch_into(frames[:12], bufs)
it = nsc.WgpuFrameInterpolator()
it.interpolate_py(frames[0], frames[1], w, h)
bs = bb = bi = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    for i in range(12):
        u.upscale_into(frames[i], out)
    bs = min(bs, (time.perf_counter() - t0) / 12)
    t0 = time.perf_counter()
    for _ in range(2):
        u.upscale_batch_into(frames[:12], bufs)
    bb = min(bb, (time.perf_counter() - t0) / 24)
    t0 = time.perf_counter()
    for i in range(12):
        it.interpolate_py(frames[i], frames[i + 1], w, h, time_t=0.5)
    bi = min(bi, (time.perf_counter() - t0) / 12)
print(f"NUS_COPY_THREADS={os.environ.get('NUS_COPY_THREADS', '-'):2s}: upscale() {bs*1e3:.3f}  upscale_batch {bb*1e3:.3f} ms/frame  interpolate_py {bi*1e3:.3f} ms/pair")
