#!/bin/bash
# one --pmc pass: VALU instruction count + busy cycles of the Lanczos x2 kernel (optionally another library via NUS_LIB_PATH)
pat=${1:-gradient}; tag=${2:-main}
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/sqv_${tag}_$pat; mkdir -p $out
[ -n "$NUS_LIB_PATH" ] && export NUS_LIB_PATH=$root/$NUS_LIB_PATH
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace -d $out --output-format csv -- python3 $root/tools/lanczos_only.py 64 2 $pat > $out.log 2>&1 || tail -3 $out.log
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lanczos3_x2<" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("$tag $pat", {k: round(tot[k]/n[k]) for k in sorted(tot)})
PY
