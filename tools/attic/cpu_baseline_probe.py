#!/usr/bin/env python
"""Where the all-cores CPU baseline of the unit spends its time (dev tool): each oracle call alone and in sequence, 10 runs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import oracle

oracle.build()
w, h = 1920, 1080
a, b = oracle.gen_gradient(w, h, 0), oracle.gen_gradient(w, h, 1)
print("omp max threads", oracle.max_threads(), "affinity", len(os.sched_getaffinity(0)))


def med(fn, n=10):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[n // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3


for th in (0, 64, 16):
    print(f"threads={th}")
    print("  warp_blend   %.2f ms (min %.2f max %.2f)" % med(lambda: oracle.warp_blend(a, b, None, 0.5, threads=th)))
    print("  lanczos3     %.2f ms (min %.2f max %.2f)" % med(lambda: oracle.lanczos3(a, 2 * w, 2 * h, threads=th)))
    mid = oracle.warp_blend(a, b, None, 0.5, threads=th)
    print("  lanczos3 mid %.2f ms (min %.2f max %.2f)" % med(lambda: oracle.lanczos3(mid, 2 * w, 2 * h, threads=th)))

    def unit():
        m = oracle.warp_blend(a, b, None, 0.5, threads=th)
        oracle.lanczos3(a, 2 * w, 2 * h, threads=th)
        oracle.lanczos3(m, 2 * w, 2 * h, threads=th)

    print("  unit         %.2f ms (min %.2f max %.2f)" % med(unit))
