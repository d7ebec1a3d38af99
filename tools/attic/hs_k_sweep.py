#!/usr/bin/env python
"""One launch of the Horn-Schunck tile kernel per K = 1..8 at a given size (dev tool; run under
rocprofv3 --kernel-trace --stats to read the per-K kernel durations)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import nu_scaler_amd as nsc

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 270)
rng = np.random.default_rng(1)
i1 = rng.random((h, w, 4), dtype=np.float32)
i2 = rng.random((h, w, 4), dtype=np.float32)
fe = nsc.FlowEstimator()
for rep in range(3):
    for k in range(1, 9):
        fe.horn_schunck(i1, i2, None, iterations=k)
print("done")
