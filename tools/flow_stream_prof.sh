#!/bin/bash
# per-kernel breakdown of the batched flow stream (rocprofv3 --stats of tools/flow_stream_bench.py); run on the GPU box
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/flowsprof
rm -rf $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o fl -- python3 $root/tools/flow_stream_bench.py "$@" > $out.log 2>&1 || tail -3 $out.log
grep "flow stream" $out.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "nus::" in n or "rocclr" in n:
        short = (re.search(r"k_\w+(<[^>]*>)?", n) or re.search(r".*", n)).group(0) if "rocclr" not in n else n
        print(f"{short:36s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {float(r['Percentage']):5.1f} %")
PY
# the Jacobi kernel per pyramid level (told apart by the launch grid)
python3 - "$(find $out -name "*kernel_trace.csv" | head -1)" <<'PY'
import csv, re, sys, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_hs_stream" in n or "k_hs_tiled" in n:
        short = re.search(r"k_\w+(<[^>]*>)?", n).group(0)
        g[(short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in g.items():
    print(f"{k[0]:28s} grid {k[1]:>7s} x {k[2]:>5s} x {k[3]:>3s}  calls {len(v):3d}  avg {sum(v)/len(v)/1e3:9.2f} us  total {sum(v)/1e6:8.2f} ms")
PY
