#!/usr/bin/env python
"""The pipeline step four ways in ONE process, interleaved rounds (dev tool): three stages (blend, upscale, upscale), the fused
pair (upscale + upscale-with-blend-on-load), and the one-launch unit kernel in both wave orders.  Outputs of every schedule are
compared with the three-stage ones (bit for bit).  usage: unit_bench.py [--frames N] [--rounds R] [--reps K] [--th T]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=8, help="steps per timing sample")
    ap.add_argument("--th", type=int, default=0)
    ap.add_argument("--patterns", default="gradient,noise")
    ap.add_argument("--no-mid", action="store_true", help="unit kernel without the in-between frame output")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    w, h, n = 1920, 1080, a.frames
    pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
    if a.th:
        pipe.upscaler.set_option("rows_per_wave", a.th)
    mid, up_real, up_mid = pipe.alloc(n, dev)
    st = torch.cuda.current_stream().cuda_stream
    unit_bytes, unit_pix = pipe.unit_bytes, pipe.unit_pixels

    def unit(order):
        def f():
            pipe.upscaler.set_option("unit_order", order)
            pipe.step_unit(frames, None if a.no_mid else mid, up_real, up_mid, st)
        return f

    sched = {"three_stage": lambda: pipe.step(frames, mid, up_real, up_mid, st),
             "fused_pair": lambda: pipe.step_fused(frames, up_real, up_mid, st),
             "unit_rbmajor": unit(1), "unit_framemajor": unit(0)}
    for pattern in a.patterns.split(","):
        frames = (syn.gradient_stream_torch if pattern == "gradient" else syn.noise_stream_torch)(n + 1, w, h, dev)
        sig0 = None
        times = {k: [] for k in sched}
        for rnd in range(a.rounds + 1):
            for name, fn in sched.items():
                if rnd == 0:
                    for t_ in (mid, up_real, up_mid):
                        t_.zero_()
                    fn()
                    torch.cuda.synchronize()
                    sig = tuple(int(x.view(torch.int32).sum(dtype=torch.int64).item()) for x in
                                ((up_real, up_mid) if name == "fused_pair" or a.no_mid else (mid, up_real, up_mid)))
                    if name == "three_stage":
                        sig0 = sig
                    ref = sig0[-len(sig):]
                    print(f"  {pattern:8s} {name:16s} outputs {'== three_stage' if sig == ref else '!= THREE_STAGE <<<<<<'}", flush=True)
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / a.reps)
        for name in sched:
            ms = statistics.median(times[name])
            print(f"{pattern:8s} {name:16s} median {ms:7.3f} ms/step  min {min(times[name]):7.3f}  = {ms * 1e3 / n:6.2f} us/unit  "
                  f"{n * unit_pix / ms / 1e3:10.0f} Mpix/s  {n * unit_bytes / ms / 1e9 * 1e3:6.0f} GB/s algorithmic "
                  f"= {n * unit_bytes / ms / 1e6 / 8000 * 100:5.1f} % of 8 TB/s", flush=True)
        del frames


if __name__ == "__main__":
    main()
