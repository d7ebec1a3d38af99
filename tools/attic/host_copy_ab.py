#!/usr/bin/env python
"""A/B of the copy pool's piece copy (NUS_COPY_STREAMING=0/1, read once per process): single upscale() calls and
upscale_batch on pageable caller buffers, Lanczos-3 1080p -> 4K.  Run once per setting, alternating."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn


def main():
    w, h = 1920, 1080
    frames = [syn.gradient_frame(w, h, k).tobytes() for k in range(12)]
    u = nsc.PyWgpuUpscaler("quality", "lanczos3")
    u.initialize(w, h, 2 * w, 2 * h)
    out = bytearray(u.output_size)
    bufs = [bytearray(u.output_size) for _ in frames]
    u.upscale_into(frames[0], out)
    u.upscale_batch_into(frames, bufs)
    best_single, best_batch = 1e9, 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for i in range(12):
            u.upscale_into(frames[i], out)
        best_single = min(best_single, (time.perf_counter() - t0) / 12)
        t0 = time.perf_counter()
        for _ in range(3):
            u.upscale_batch_into(frames, bufs)
        best_batch = min(best_batch, (time.perf_counter() - t0) / 36)
    print(f"NUS_COPY_STREAMING={os.environ.get('NUS_COPY_STREAMING', '-')} NUS_COPY_THREADS={os.environ.get('NUS_COPY_THREADS', '-')}: "
          f"upscale() {best_single*1e3:6.3f} ms   upscale_batch(12) {best_batch*1e3:6.3f} ms/frame (best of 4)")


if __name__ == "__main__":
    main()
