#!/usr/bin/env python
"""Durations of the Jacobi kernel's launches in launch order (dev tool): after tools/flow_stream_prof.sh, reads its kernel trace."""
import csv
import glob
import sys

f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/flowsprof") + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_hs_stream" in r["Kernel_Name"] or "k_hs_tiled" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(" ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in rows))
