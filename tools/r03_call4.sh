#!/bin/bash
# round 3, call 4: dense-flow warp kernel (EXACT / FMA, 4 px per lane): parity, timing; NUMA placement of the host path
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call4
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_flow.py -x -q -m gpu -k "warp or interp or flow or fused or motion or bgra" > $out/warp_tests.txt 2>&1; rc=$?; echo "warp tests rc=$rc"; tail -8 $out/warp_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/warp_bench.py > $out/warp_bench.txt 2>&1; echo "rc=$?"; grep -v amdgpu $out/warp_bench.txt
timeout -k 10 600 bash tools/numa_probe.sh > $out/numa_probe.txt 2>&1; echo "rc=$?"; cat $out/numa_probe.txt
