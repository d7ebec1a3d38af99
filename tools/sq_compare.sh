#!/bin/bash
# SQ counters of k_lanczos3_x2 for several builds of the library on one box: tools/sq_compare.sh <pattern> name=lib.so ...
pat=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name="${spec%%=*}"; lib="${spec#*=}"
  out=$root/gpurun_out/sqc_${name}_$pat; rm -rf $out; mkdir -p $out
  export NUS_LIB_PATH=$root/$lib
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" \
             "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_WR" \
             "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_LDS SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $out/p$i --output-format csv -- python3 $root/tools/lanczos_only.py 64 2 $pat > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
  done
  echo "== $name ($pat)"
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lanczos3_x2I" in r["Kernel_Name"] and "edges" not in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:34s} {tot[k]/n[k]:16.0f}  per launch ({n[k]} launches)")
PY
done
