#!/usr/bin/env python
"""A/B timing of Lanczos x2 kernel variants in ONE process, interleaved rounds (dev tool).
usage: lz_variants.py [--frames N] [--rounds R] [--th T] name=lib.so ...
Every library is loaded side by side through the C ABI; each round runs every variant once on the same
frames; outputs are compared with the first variant's (bit for bit).  Prints median / min us per frame."""
import argparse
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nu_scaler_amd import synthetic as syn

vp, u32, i64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int64


class Lib:
    def __init__(self, path, th, alg=2, dims=(1920, 1080, 3840, 2160), exact=False):
        L = ctypes.CDLL(os.path.abspath(path))
        L.nus_upscaler_create.restype = vp
        L.nus_upscaler_create.argtypes = [ctypes.c_int, ctypes.c_int]
        L.nus_upscaler_initialize.argtypes = [vp, u32, u32, u32, u32]
        L.nus_upscaler_set_option.argtypes = [vp, ctypes.c_char_p, i64]
        L.nus_upscaler_upscale_device.argtypes = [vp, vp, vp, u32, vp]
        L.nus_upscaler_set_profiling.argtypes = [vp, ctypes.c_int]
        L.nus_upscaler_profile_collect.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]
        L.nus_upscaler_last_error.restype = ctypes.c_char_p
        L.nus_upscaler_last_error.argtypes = [vp]
        self.L = L
        self.h = L.nus_upscaler_create(alg, 2)
        if th:
            assert L.nus_upscaler_set_option(self.h, b"rows_per_wave", th) == 0
        if exact:
            L.nus_upscaler_set_lanczos_mode.argtypes = [vp, ctypes.c_int]
            assert L.nus_upscaler_set_lanczos_mode(self.h, 1) == 0
        assert L.nus_upscaler_initialize(self.h, *dims) == 0, L.nus_upscaler_last_error(self.h)
        L.nus_upscaler_set_profiling(self.h, 1)

    def run(self, src, dst, n, stream):
        rc = self.L.nus_upscaler_upscale_device(self.h, src, dst, n, stream)
        assert rc == 0, self.L.nus_upscaler_last_error(self.h)

    def collect(self):
        nl, ms = ctypes.c_uint64(), ctypes.c_double()
        self.L.nus_upscaler_profile_collect(self.h, ctypes.byref(nl), ctypes.byref(ms))
        return ms.value / max(nl.value, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--th", type=int, default=0)
    ap.add_argument("--alg", default="lanczos3", help="nearest, bilinear, lanczos3, bicubic, triangle")
    ap.add_argument("--dims", default="1920x1080:3840x2160")
    ap.add_argument("--patterns", default="gradient,noise")
    ap.add_argument("--exact", action="store_true", help="the resize filters' EXACT mode (separate multiply and add, as the CPU)")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = a.frames
    alg = {"nearest": 0, "bilinear": 1, "lanczos3": 2, "bicubic": 3, "triangle": 4}[a.alg]
    (iw, ih), (ow, oh) = [tuple(int(v) for v in part.split("x")) for part in a.dims.split(":")]
    libs = [(s.split("=")[0], Lib(s.split("=")[1], a.th, alg, (iw, ih, ow, oh), a.exact)) for s in a.libs]
    out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
    mb = (iw * ih + ow * oh) * 4 / 1e6
    st = torch.cuda.current_stream().cuda_stream
    print(f"{a.alg} {iw}x{ih} -> {ow}x{oh}, {n} frames per launch, {mb:.2f} MB per frame algorithmic")
    for pattern in a.patterns.split(","):
        # gradient: opaque; noise: every alpha its own; alpha128: the gradient with alpha 128 everywhere; alpharegions: thirds of
        # alpha 0 / 128 / 255 (flat alphas take the 3-channel path since round 5)
        frames = (syn.noise_stream_torch if pattern == "noise" else syn.gradient_stream_torch)(n, iw, ih, dev)
        if pattern == "alpha128":
            frames[..., 3] = 128
        elif pattern == "alpharegions":
            frames[:, :ih // 3, :, 3] = 0
            frames[:, ih // 3:2 * ih // 3, :, 3] = 128
        ref = None
        times = {name: [] for name, _ in libs}
        for rnd in range(a.rounds + 1):
            for name, lib in libs:
                if rnd == 0:
                    out.zero_()
                lib.run(frames.data_ptr(), out.data_ptr(), n, st)
                torch.cuda.synchronize()
                ms = lib.collect()
                if rnd == 0:  # warm-up round doubles as the equality check
                    sig = (int(out.view(torch.int32).sum(dtype=torch.int64).item()), int(out[n // 2].to(torch.int64).pow(2).sum().item()))
                    if ref is None:
                        ref = sig
                    print(f"  {pattern:8s} {name:14s} output {'== first variant' if sig == ref else '!= FIRST VARIANT  <<<<<<'}", flush=True)
                else:
                    times[name].append(ms * 1e3 / n)
        for name, _ in libs:
            t = times[name]
            med = statistics.median(t)
            print(f"{pattern:8s} {name:14s} median {med:6.2f} us/frame  min {min(t):6.2f}  max {max(t):6.2f}   "
                  f"{mb / med:5.2f} TB/s = {mb / med / 8 * 100:4.1f} % of 8 TB/s", flush=True)
        del frames


if __name__ == "__main__":
    main()
