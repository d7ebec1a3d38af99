#!/bin/bash
# round 3, call 5: warp kernel again (exact blend), unit kernel A/B (one DMA in the real-frame role), NUMA placement of the host path
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call5
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_flow.py -x -q -m gpu -k "warp or interp or flow or fused or motion or bgra" > $out/warp_tests.txt 2>&1; rc=$?; echo "warp tests rc=$rc"; tail -4 $out/warp_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/warp_bench.py > $out/warp_bench.txt 2>&1; echo "rc=$?"; grep -v amdgpu $out/warp_bench.txt
for v in product onedma product onedma; do
  echo "== unit kernel: $v"
  if [ $v = onedma ]; then export NUS_LIB_PATH=$root/tools/_ablate/lib_onedma.so; else unset NUS_LIB_PATH; fi
  timeout -k 10 300 python3 tools/unit_bench.py --rounds 3 2>&1 | grep -E "unit_rbmajor|three_stage " | grep median
done > $out/unit_one_dma_ab.txt 2>&1; unset NUS_LIB_PATH; cat $out/unit_one_dma_ab.txt
timeout -k 10 600 bash tools/numa_probe.sh > $out/numa_probe.txt 2>&1; echo "rc=$?"; cat $out/numa_probe.txt
