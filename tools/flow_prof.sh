#!/bin/bash
# per-kernel breakdown of the flow front end (rocprofv3 --stats of tools/flow_bench.py); run on the GPU box
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/flowprof
rm -rf $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o fl -- python3 $root/tools/flow_bench.py "$@" > $out.log 2>&1 || tail -3 $out.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "nus::" in n or "rocclr" in n:
        short = (re.search(r"k_\w+(<[^>]*>)?", n) or re.search(r".*", n)).group(0) if "rocclr" not in n else n
        print(f"{short:36s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:8.2f} us  {float(r['Percentage']):5.1f} %")
PY
