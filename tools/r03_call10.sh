#!/bin/bash
# round 3, call 10: k_resize_down with the LDS-DMA row ring: parity, then depth A/B on the common down-scales; pin API test
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call10
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resize or down or batch or general or bicubic or triangle" > $out/down_tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $out/down_tests.txt
[ $rc -eq 0 ] || exit 1
for v in dd0 product dd2 dd8 dd0 product; do
  if [ $v = product ]; then unset NUS_LIB_PATH; else export NUS_LIB_PATH=$root/tools/_ablate/lib_$v.so; fi
  for dims in "3840 2160 1920 1080" "3840 2160 1280 720" "2560 1440 1920 1080" "1920 1080 1280 720"; do
    for pat in gradient noise; do
      echo -n "$v $pat $dims: "; NUS_PATTERN=$pat timeout -k 10 120 python3 tools/general_bench.py $dims 32 2>&1 | grep "us/frame" | grep -i lanczos | head -1
    done
  done
done > $out/resize_down_depth_ab.txt 2>&1; unset NUS_LIB_PATH; cat $out/resize_down_depth_ab.txt
