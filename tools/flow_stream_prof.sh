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
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "nus::" in n or "rocclr" in n:
        short = n.split("::")[-1].split("(")[0] if "rocclr" not in n else n
        print(f"{short:36s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {float(r['Percentage']):5.1f} %")
PY
