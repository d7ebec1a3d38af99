#!/bin/bash
# SQ counter passes for one kernel name substring: tools/sq_kernel.sh <substr> -- <python script args...>
sub=$1; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/sqk_$sub; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_WR" \
           "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i --output-format csv -- python3 "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$sub" in r["Kernel_Name"] and "edges" not in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:34s} {tot[k]/n[k]:16.0f}  per launch ({n[k]} launches)")
PY
