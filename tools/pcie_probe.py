#!/usr/bin/env python
"""Pinned host <-> device copy ceilings of this box for the frame sizes of the host path (dev tool; the product's own
figure is bench.py's config.host_path.pcie_ceiling): hipMemcpyAsync of one 1080p input frame (8.3 MB) and one 4K
output frame (33.2 MB) between pinned host memory and HBM, each direction alone and both at once on two streams;
plus what one thread gets from memcpy between two pageable 33 MB buffers."""
import time

import numpy as np
import torch


def main():
    dev = torch.device("cuda:0")
    sizes = {"1080p_in": 1920 * 1080 * 4, "4k_out": 3840 * 2160 * 4}
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, n in sizes.items():
        h_a = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        h_b = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        d_a = torch.empty(n, dtype=torch.uint8, device=dev)
        d_b = torch.empty(n, dtype=torch.uint8, device=dev)
        reps = 40

        def run(h2d, d2h):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                if h2d:
                    with torch.cuda.stream(s1):
                        d_a.copy_(h_a, non_blocking=True)
                if d2h:
                    with torch.cuda.stream(s2):
                        h_b.copy_(d_b, non_blocking=True)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        run(True, True)
        t_h2d, t_d2h, t_both = run(True, False), run(False, True), run(True, True)
        print(f"{name:9s} {n/1e6:6.2f} MB  H2D {n/t_h2d/1e9:6.2f} GB/s ({t_h2d*1e3:.3f} ms)  D2H {n/t_d2h/1e9:6.2f} GB/s "
              f"({t_d2h*1e3:.3f} ms)  both at once {t_both*1e3:.3f} ms per pair = {2*n/t_both/1e9:6.2f} GB/s summed", flush=True)
    a = np.random.randint(0, 255, sizes["4k_out"], dtype=np.uint8)
    b = np.empty_like(a)
    np.copyto(b, a)
    t0 = time.perf_counter()
    for _ in range(10):
        np.copyto(b, a)
    dt = (time.perf_counter() - t0) / 10
    print(f"one-thread memcpy of 33.2 MB pageable -> pageable: {a.nbytes/dt/1e9:.1f} GB/s ({dt*1e3:.2f} ms)")
    import os
    print("cpus in affinity mask:", len(os.sched_getaffinity(0)), " os.cpu_count:", os.cpu_count())


if __name__ == "__main__":
    main()
