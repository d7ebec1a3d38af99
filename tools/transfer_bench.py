#!/usr/bin/env python
"""nus_download / nus_upload rates (dev tool): a 1080p frame, a 4K frame and 8 4K frames between HBM and (a) a resident pageable numpy
array, (b) a pinned torch tensor (direct DMA: the PCIe ceiling), median of 9 after 2 warm-ups."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nu_scaler_amd import _capi, transfer

L = _capi.lib()
s = torch.cuda.current_stream().cuda_stream
print(f"copy pool workers: NUS_COPY_THREADS={os.environ.get('NUS_COPY_THREADS', '(default)')}")
for name, n in (("1080p frame", 1920 * 1080 * 4), ("4K frame", 3840 * 2160 * 4), ("8 x 4K frames", 8 * 3840 * 2160 * 4)):
    d = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda:0")
    host = np.zeros(n, np.uint8)
    pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    torch.cuda.synchronize()

    def med(fn):
        for _ in range(2):
            fn()
        v = []
        for _ in range(9):
            t0 = time.perf_counter()
            fn()
            v.append(time.perf_counter() - t0)
        v.sort()
        return v[4]

    t_dp = med(lambda: L.nus_download(host.ctypes.data, d.data_ptr(), n, s))
    t_dd = med(lambda: L.nus_download(pinned.data_ptr(), d.data_ptr(), n, s))

    def up_pageable():
        L.nus_upload(d.data_ptr(), host.ctypes.data, n, s)
        torch.cuda.synchronize()

    t_up = med(up_pageable)
    t_ud = med(lambda: L.nus_upload(d.data_ptr(), pinned.data_ptr(), n, s))
    assert np.array_equal(host, pinned.numpy())
    print(f"{name:14s} {n / 1e6:8.2f} MB   download: pageable {t_dp * 1e3:7.3f} ms = {n / t_dp / 1e9:5.1f} GB/s, pinned (direct DMA) {t_dd * 1e3:7.3f} ms = "
          f"{n / t_dd / 1e9:5.1f} GB/s   upload: pageable {t_up * 1e3:7.3f} ms = {n / t_up / 1e9:5.1f} GB/s, pinned {t_ud * 1e3:7.3f} ms = {n / t_ud / 1e9:5.1f} GB/s",
          flush=True)
