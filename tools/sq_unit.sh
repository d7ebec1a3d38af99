#!/bin/bash
# SQ counter passes over the one-launch pipeline step (k_lanczos3_x2<.., UNIT>), gradient and noise: tools/sq_unit.sh
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for pat in gradient noise; do
  out=$root/gpurun_out/sq_unit_$pat; rm -rf $out; mkdir -p $out
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
             "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH" \
             "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" \
             "SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $out/p$i --output-format csv -- python3 $root/tools/unit_only.py 64 2 $pat > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
  done
  echo "## $pat"
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter(); dur = []
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lanczos3_x2<" in r["Kernel_Name"] and ", true, " in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for f in glob.glob("$out/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lanczos3_x2<" in r["Kernel_Name"] and ", true, " in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(tot): print(f"{k:34s} {tot[k]/n[k]:16.0f}  per launch ({n[k]} launches)")
if dur: print(f"kernel duration (pass 1)           {sum(dur)/len(dur)/1e3:16.1f}  us per launch of 64 units")
PY
done
