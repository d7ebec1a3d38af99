#!/bin/bash
# round 3, call 12: full GPU suite after the request-after-read guard in the x2 kernel and the resize_down split; A/B of the guard
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call12
rm -rf $out && mkdir -p $out
cd $root
export NUS_EXPECT_GPU=1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -4 $out/gpu_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/lz_variants.py --rounds 5 cur=nu_scaler_amd/lib/libnuscaler_hip.so nodep=tools/_ablate/lib_nodep.so > $out/x2_request_guard_ab.txt 2>&1; grep -v amdgpu $out/x2_request_guard_ab.txt
for v in product nodep product nodep; do
  echo "== unit step: $v"
  if [ $v = nodep ]; then export NUS_LIB_PATH=$root/tools/_ablate/lib_nodep.so; else unset NUS_LIB_PATH; fi
  timeout -k 10 300 python3 tools/unit_bench.py --rounds 3 2>&1 | grep -E "unit_rbmajor|three_stage " | grep median
done > $out/unit_request_guard_ab.txt 2>&1; unset NUS_LIB_PATH; cat $out/unit_request_guard_ab.txt
