#!/usr/bin/env python
"""Pageable host path, Lanczos-3 1080p -> 4K: ms per frame by how the output frame is cut into D2H pieces
(options single_out_plan for upscale(), batch_out_chunks for upscale_batch and the stream ring)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

w, h = 1920, 1080
frames = [syn.gradient_frame(w, h, k).tobytes() for k in range(12)]


def make(**opts):
    u = nsc.PyWgpuUpscaler("quality", "lanczos3")
    for k, v in opts.items():
        u.set_option(k, v)
    u.initialize(w, h, 2 * w, 2 * h)
    return u


for rep in range(2):
    for alg in ("lanczos3", "bilinear"):
        line = []
        for bands in (0, 1):
            u = nsc.PyWgpuUpscaler("quality", alg)
            u.set_option("single_bands", bands)
            u.initialize(w, h, 2 * w, 2 * h)
            out = bytearray(u.output_size)
            u.upscale_into(frames[0], out)
            best, ts = 1e9, []
            for _ in range(4):
                t0 = time.perf_counter()
                for i in range(12):
                    u.upscale_into(frames[i], out)
                best = min(best, (time.perf_counter() - t0) / 12)
            line.append(f"single_bands {bands}: {best*1e3:.3f}")
        print(f"upscale() {alg:9s} " + "   ".join(line), flush=True)
    line = []
    for plan in (0, 1, 2, 3):
        u = make(single_out_plan=plan)
        out = bytearray(u.output_size)
        u.upscale_into(frames[0], out)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            for i in range(12):
                u.upscale_into(frames[i], out)
            best = min(best, (time.perf_counter() - t0) / 12)
        line.append(f"plan {plan}: {best*1e3:.3f}")
    print("upscale()        " + "   ".join(line), flush=True)
    line = []
    for chunks in (1, 2, 3, 4, 8):
        u = make(batch_out_chunks=chunks)
        bufs = [bytearray(u.output_size) for _ in frames]
        u.upscale_batch_into(frames, bufs)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(3):
                u.upscale_batch_into(frames, bufs)
            best = min(best, (time.perf_counter() - t0) / 36)
        line.append(f"{chunks} pieces: {best*1e3:.3f}")
    print("upscale_batch(12) " + "   ".join(line), flush=True)
