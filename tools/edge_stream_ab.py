#!/usr/bin/env python
"""The x2 edge-column pass behind the main kernel (option edge_stream 0) or beside it on the upscaler's second stream (1): the
one-launch step and the plain upscale launch, interleaved rounds in ONE process, outputs compared (dev tool, round 4)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nu_scaler_amd as nsc
from nu_scaler_amd import synthetic as syn

dev = torch.device("cuda:0")
w, h, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 300
rounds, reps = 7, 8
pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
mid, up_real, up_mid = pipe.alloc(n, dev)
st = torch.cuda.current_stream().cuda_stream
for pattern in ("gradient", "noise"):
    frames = (syn.gradient_stream_torch if pattern == "gradient" else syn.noise_stream_torch)(n + 1, w, h, dev)
    legs = {"unit": lambda: pipe.step_unit(frames, mid, up_real, up_mid, st),
            "plain": lambda: pipe.upscaler.upscale_device(frames.data_ptr(), up_real.data_ptr(), n, st)}
    times = {(m, k): [] for m in (0, 1) for k in legs}
    sigs = {}
    for rnd in range(rounds + 1):
        for m in (0, 1):
            pipe.upscaler.set_option("edge_stream", m)
            for k, fn in legs.items():
                if rnd == 0:
                    for t_ in (mid, up_real, up_mid):
                        t_.zero_()
                    fn()
                    torch.cuda.synchronize()
                    sigs[(m, k)] = tuple(int(x.view(torch.int32).sum(dtype=torch.int64).item()) for x in (mid, up_real, up_mid))
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[(m, k)].append(e0.elapsed_time(e1) / reps)
    for k in legs:
        same = sigs[(0, k)] == sigs[(1, k)]
        a, b = statistics.median(times[(0, k)]), statistics.median(times[(1, k)])
        print(f"{pattern:8s} {k:5s} edge pass behind {a:7.3f} ms  beside {b:7.3f} ms per {n}-frame launch  ({(b / a - 1) * 100:+5.1f} %)  outputs "
              f"{'identical' if same else 'DIFFER <<<<'}", flush=True)
    del frames
