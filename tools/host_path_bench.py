#!/usr/bin/env python
"""End-to-end rate through the trait-shaped host API: host bytes in -> host bytes out
(PCIe + staging included).  This is number (ii) of BASELINE.md section 3; bench.py reports
number (i), the HBM-resident rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn


def main():
    w, h = 1920, 1080
    frames = [syn.gradient_frame(w, h, k).tobytes() for k in range(12)]
    for alg in ("nearest", "bilinear", "lanczos3"):
        u = nsc.PyWgpuUpscaler("quality", alg)
        u.initialize(w, h, 2 * w, 2 * h)
        out = bytearray(u.output_size)
        u.upscale_into(frames[0], out)
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            u.upscale_into(frames[i % len(frames)], out)
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for i in range(6):
            o = u.upscale(frames[i])  # returns a fresh bytes object, like the reference's PyBytes
        dtp = (time.perf_counter() - t0) / 6
        del o
        outs = u.upscale_batch(frames)  # first call allocates the 3 pinned slots
        del outs
        t0 = time.perf_counter()
        outs = u.upscale_batch(frames)
        dtb = (time.perf_counter() - t0) / len(frames)
        del outs
        bufs = [bytearray(u.output_size) for _ in frames]
        u.upscale_batch_into(frames, bufs)
        t0 = time.perf_counter()
        for _ in range(3):
            u.upscale_batch_into(frames, bufs)
        dti = (time.perf_counter() - t0) / (3 * len(frames))
        del bufs
        print(f"{alg:9s} upscale_batch_into(12, caller buffers): {dti*1e3:7.3f} ms/frame = {1/dti:7.1f} frames/s = {dt/dti:4.2f}x the single-call rate")
        print(f"{alg:9s} upscale(): {dt*1e3:7.3f} ms/frame = {1/dt:7.1f} frames/s, {(u.input_size+u.output_size)/dt/1e9:6.2f} GB/s host<->device;"
              f" upscale()->bytes {dtp*1e3:6.2f} ms; upscale_batch(12): {dtb*1e3:7.3f} ms/frame = {1/dtb:7.1f} frames/s; kernel {u.get_last_gpu_duration_ms()*1e3:7.1f} us")
    it = nsc.WgpuFrameInterpolator()
    it.interpolate_py(frames[0], frames[1], w, h)
    t0 = time.perf_counter()
    for i in range(10):
        it.interpolate_py(frames[i], frames[i + 1], w, h, time_t=0.5)
    dt = (time.perf_counter() - t0) / 10
    print(f"interpolate_py(): {dt*1e3:7.3f} ms/pair = {1/dt:7.1f} pairs/s; kernel {it.get_last_gpu_duration_ms()*1e3:7.1f} us")
    # full unit through the host API: interp + 2 upscales (the 60 frames/s/GPU target is judged here)
    u = nsc.PyWgpuUpscaler("quality", "lanczos3")
    u.initialize(w, h, 2 * w, 2 * h)
    out = bytearray(u.output_size)
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        mid = it.interpolate_py(frames[i], frames[i + 1], w, h, time_t=0.5)
        u.upscale_into(frames[i], out)
        u.upscale_into(mid, out)
    dt = (time.perf_counter() - t0) / n
    print(f"host-path unit (interp + 2x Lanczos-3 to 4K): {dt*1e3:7.3f} ms/unit = {1/dt:6.1f} source frames/s = {2/dt:6.1f} 4K output frames/s")


if __name__ == "__main__":
    main()
