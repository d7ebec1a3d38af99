#!/bin/bash
# Round 6: non-temporal stores / loads in the FAST flow front end (timing-only A/B libraries built with
#   UNIT=nus_k_flow tools/build_lz_variants.sh pyrnt="-DNUS_PYR_FAST_NT=1" hsnt1="-DNUS_HS_FAST_NT=1" hsnt3="-DNUS_HS_FAST_NT=3" allnt="-DNUS_PYR_FAST_NT=1 -DNUS_HS_FAST_NT=3")
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/flow_nt_ab.txt
{
for r in 1 2; do
  for v in product pyrnt hsnt1 hsnt3 allnt; do
    lib=""; [ $v != product ] && lib=tools/_ablate/lib_$v.so
    echo "== $v (round $r)"
    NUS_LIB_PATH=$lib timeout -k 10 200 python3 tools/flow_stream_bench.py 101 9 2>&1 | grep "flow stream" | tail -1
    NUS_LIB_PATH=$lib timeout -k 10 200 python3 tools/motion_default.py 2>&1 | grep "Rg16Float"
  done
done
} > $out 2>&1
cat $out
