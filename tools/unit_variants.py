#!/usr/bin/env python
"""A/B timing of library variants on the ONE-LAUNCH pipeline step (nus_upscaler_upscale_unit_device) in ONE process, interleaved
rounds (dev tool; the UNIT counterpart of lz_variants.py).  usage: unit_variants.py [--frames N] [--rounds R] [--reps K]
[--patterns gradient,noise] name=lib.so ...   Every library is loaded side by side through the C ABI; each round runs every
variant `reps` times between two events on the launch stream; outputs are compared with the first variant's (the timing-only
ablation builds of tools/build_lz_variants.sh write wrong pixels by design: their lines say so)."""
import argparse
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nu_scaler_amd import synthetic as syn

vp, u32, i64, sz, f32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float


class Lib:
    def __init__(self, path, th, w, h):
        L = ctypes.CDLL(os.path.abspath(path))
        L.nus_upscaler_create.restype = vp
        L.nus_upscaler_create.argtypes = [ctypes.c_int, ctypes.c_int]
        L.nus_upscaler_initialize.argtypes = [vp, u32, u32, u32, u32]
        L.nus_upscaler_set_option.argtypes = [vp, ctypes.c_char_p, i64]
        L.nus_upscaler_upscale_unit_device.argtypes = [vp, vp, sz, vp, sz, f32, vp, vp, vp, u32, vp]
        L.nus_upscaler_upscale_device.argtypes = [vp, vp, vp, u32, vp]
        L.nus_upscaler_last_error.restype = ctypes.c_char_p
        L.nus_upscaler_last_error.argtypes = [vp]
        self.L = L
        self.h = L.nus_upscaler_create(2, 2)
        if th:
            assert L.nus_upscaler_set_option(self.h, b"rows_per_wave", th) == 0
        assert L.nus_upscaler_initialize(self.h, w, h, 2 * w, 2 * h) == 0, L.nus_upscaler_last_error(self.h)

    def unit(self, frames, fb, mid, up_real, up_mid, n, stream):
        rc = self.L.nus_upscaler_upscale_unit_device(self.h, frames, fb, frames + fb, fb, 0.5, mid, up_real, up_mid, n, stream)
        assert rc == 0, self.L.nus_upscaler_last_error(self.h)

    def plain(self, frames, up_real, n, stream):
        rc = self.L.nus_upscaler_upscale_device(self.h, frames, up_real, n, stream)
        assert rc == 0, self.L.nus_upscaler_last_error(self.h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--th", type=int, default=0)
    ap.add_argument("--patterns", default="gradient,noise")
    ap.add_argument("--plain", action="store_true", help="also time the plain upscale launch (k_lanczos3_x2 alone) per variant")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    w, h, n = 1920, 1080, a.frames
    fb = w * h * 4
    libs = [(s.split("=")[0], Lib(s.split("=")[1], a.th, w, h)) for s in a.libs]
    mid = torch.empty((n, h, w, 4), dtype=torch.uint8, device=dev)
    up_real = torch.empty((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
    up_mid = torch.empty_like(up_real)
    unit_mb = (3 * fb + 2 * 5 * fb) / 1e6
    st = torch.cuda.current_stream().cuda_stream
    print(f"one-launch step, {n} units per launch, {unit_mb:.2f} MB per unit algorithmic; {a.reps} launches per sample, {a.rounds} rounds")
    for pattern in a.patterns.split(","):
        frames = (syn.gradient_stream_torch if pattern == "gradient" else syn.noise_stream_torch)(n + 1, w, h, dev)
        ref = None
        modes = ["unit"] + (["plain"] if a.plain else [])
        times = {(name, m): [] for name, _ in libs for m in modes}
        for rnd in range(a.rounds + 1):
            for name, lib in libs:
                for m in modes:
                    run = (lambda: lib.unit(frames.data_ptr(), fb, mid.data_ptr(), up_real.data_ptr(), up_mid.data_ptr(), n, st)) \
                        if m == "unit" else (lambda: lib.plain(frames.data_ptr(), up_real.data_ptr(), n, st))
                    if rnd == 0:
                        for t_ in (mid, up_real, up_mid):
                            t_.zero_()
                        run()
                        torch.cuda.synchronize()
                        if m == "unit":
                            sig = tuple(int(x.view(torch.int32).sum(dtype=torch.int64).item()) for x in (mid, up_real, up_mid))
                            if ref is None:
                                ref = sig
                            print(f"  {pattern:8s} {name:14s} outputs {'== first variant' if sig == ref else '!= first variant (timing-only build?)'}",
                                  flush=True)
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.reps):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    times[(name, m)].append(e0.elapsed_time(e1) / a.reps)
        for name, _ in libs:
            for m in modes:
                t = times[(name, m)]
                med = statistics.median(t)
                per = med * 1e3 / n
                mb = unit_mb if m == "unit" else 5 * fb / 1e6
                print(f"{pattern:8s} {name:14s} {m:5s} median {med:7.3f} ms per launch  min {min(t):7.3f}  max {max(t):7.3f}  = {per:6.2f} us per "
                      f"{'unit' if m == 'unit' else 'frame'}  {mb / per:5.2f} TB/s algorithmic = {mb / per / 8 * 100:4.1f} % of 8 TB/s", flush=True)
        del frames


if __name__ == "__main__":
    main()
