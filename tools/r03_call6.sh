#!/bin/bash
# round 3, call 6: warp kernel layouts A/B, host-path state probe, CPU baseline probe
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call6
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "warp" > $out/warp_tests.txt 2>&1; rc=$?; echo "warp tests rc=$rc"; tail -3 $out/warp_tests.txt
[ $rc -eq 0 ] || exit 1
for v in product wold wx1r4 wx4r1 wx2r2 wx1r2 wx4r2 product; do
  echo "== warp kernel: $v"
  if [ $v = product ]; then unset NUS_LIB_PATH; else export NUS_LIB_PATH=$root/tools/_ablate/lib_$v.so; fi
  timeout -k 10 200 python3 tools/warp_bench.py 2>&1 | grep warp_blend_flow
done > $out/warp_layouts_ab.txt 2>&1; unset NUS_LIB_PATH; cat $out/warp_layouts_ab.txt
timeout -k 10 400 python3 tools/host_path_state_probe.py > $out/host_path_state.txt 2>&1; echo "rc=$?"; grep -v amdgpu $out/host_path_state.txt
timeout -k 10 300 python3 tools/cpu_baseline_probe.py > $out/cpu_probe.txt 2>&1; echo "rc=$?"; cat $out/cpu_probe.txt
