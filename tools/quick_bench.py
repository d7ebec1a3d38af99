#!/usr/bin/env python
"""Per-kernel device-resident timing (hipEvents on the launch stream), for tuning.
usage: python tools/quick_bench.py [--frames N] [--reps R] [--sweep]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn


def timed(fn, reps, warm_seconds=1.0, rounds=5, after_warmup=None):
    """ms per call of `fn`: >= `warm_seconds` of back-to-back warm launches first (clocks and caches where a long run has them --
    one cold call and ten repetitions read 10.5 us per frame for a kernel rocprofv3 has at 8.8 over 460 calls), then the MEDIAN of
    `rounds` timed brackets of `reps` calls each.  `after_warmup()` runs between the two (resets per-kernel event collectors, so
    that "main kernel" covers the timed calls only)."""
    import time

    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_seconds:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    if after_warmup is not None:
        after_warmup()
    got = []
    for _ in range(max(1, rounds)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        got.append(a.elapsed_time(b) / reps)
    got.sort()
    return got[len(got) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10, help="calls per timed bracket")
    ap.add_argument("--rounds", type=int, default=5, help="timed brackets per case; the median is reported")
    ap.add_argument("--warm-seconds", type=float, default=1.0, help="warm launches before the timed brackets, per case")
    ap.add_argument("--pattern", default="noise")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated algorithm names; skips the interpolation legs")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    w, h, n = 1920, 1080, args.frames
    frames = (syn.noise_stream_torch(n + 1, w, h, dev) if args.pattern == "noise" else syn.gradient_stream_torch(n + 1, w, h, dev))
    s = torch.cuda.current_stream().cuda_stream
    up_bytes = (w * h + 4 * w * h) * 4
    out = torch.empty((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
    print(f"frames={n} pattern={args.pattern}  (upscale: {up_bytes/1e6:.2f} MB/frame algorithmic)")
    cases = [("nearest", {}, {}), ("nearest", {}, {"force_general": 1}), ("bilinear", {}, {}), ("bilinear", {}, {"force_general": 1}),
             ("lanczos3", {"lanczos_mode": "fma"}, {}), ("lanczos3", {"lanczos_mode": "exact"}, {}),
             ("bicubic", {"lanczos_mode": "fma"}, {}), ("easu", {}, {}), ("fsr1", {}, {}),
             ("easu", {}, {"fsr_fast": 1}), ("fsr1", {}, {"fsr_fast": 1})]
    if args.only:
        cases = [c for c in cases if c[0] in args.only.split(",")]
    if args.sweep:
        for th in (4, 8, 12, 16, 24, 32, 36, 48, 64):
            cases.append(("lanczos3", {"lanczos_mode": "fma"}, {"rows_per_wave": th}))
    for alg, kw, opts in cases:
        u = nsc.PyWgpuUpscaler("quality", alg, **kw)
        for k, v in opts.items():
            u.set_option(k, v)
        if os.environ.get("NUS_BENCH_FORMAT"):
            u.set_input_format(os.environ["NUS_BENCH_FORMAT"])
        u.initialize(w, h, 2 * w, 2 * h)
        u.set_profiling(True)
        ms = timed(lambda: u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s), args.reps, args.warm_seconds, args.rounds,
                   after_warmup=u.profile_collect)  # (collect = wait, return, RESET: the warm-up's launches are not in "main kernel")
        nl, kms = u.profile_collect()
        us = ms * 1e3 / n
        kus = kms * 1e3 / max(nl, 1) / n
        print(f"{alg:9s} {u.kernel_variant:24s} {str(kw)+str(opts):48s} {us:8.2f} us/frame (main kernel {kus:6.2f})  {up_bytes/kus/1e6:6.2f} TB/s  {100*up_bytes/kus/1e6/8.0:5.1f}% of 8 TB/s", flush=True)
    if args.only:
        return
    it = nsc.WgpuFrameInterpolator()
    mid = torch.empty((n, h, w, 4), dtype=torch.uint8, device=dev)
    fb = w * h * 4
    ms = timed(lambda: it.interpolate_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, 0, w, h, 0.5, mid.data_ptr(), n, s), args.reps, args.warm_seconds, args.rounds)
    us = ms * 1e3 / n
    print(f"{'interp':9s} {'blend_zero_flow':24s} {'':48s} {us:8.2f} us/pair   {3*fb/us/1e6:6.2f} TB/s  {100*3*fb/us/1e6/8.0:5.1f}% of 8 TB/s")
    flow = torch.zeros((n, h, w, 2), dtype=torch.float32, device=dev)
    flow[..., 0] = -1.0
    ms = timed(lambda: it.interpolate_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, flow.data_ptr(), w, h, 0.5, mid.data_ptr(), n, s), args.reps, args.warm_seconds, args.rounds)
    us = ms * 1e3 / n
    print(f"{'interp':9s} {'warp_blend_flow':24s} {'':48s} {us:8.2f} us/pair   {5*fb/us/1e6:6.2f} TB/s  {100*5*fb/us/1e6/8.0:5.1f}% of 8 TB/s")
    del flow
    pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
    mid, up_real, up_mid = pipe.alloc(n, dev)
    ms = timed(lambda: pipe.step(frames, mid, up_real, up_mid, s), args.reps, args.warm_seconds, args.rounds)
    us = ms * 1e3 / n
    msf = timed(lambda: pipe.step_fused(frames, up_real, up_mid, s), args.reps, args.warm_seconds, args.rounds)
    usf = msf * 1e3 / n
    print(f"fused unit (lanczos + blend-in-load lanczos): {usf:8.2f} us/unit  {pipe.unit_pixels/usf:8.1f} Mpix/s (BASELINE unit pixels)")
    print(f"pipeline unit (interp + 2x lanczos): {us:8.2f} us/unit  {pipe.unit_pixels/us:8.1f} Mpix/s  {pipe.unit_bytes/us/1e6:6.2f} TB/s  {1e6/us:8.0f} units/s")


if __name__ == "__main__":
    main()
