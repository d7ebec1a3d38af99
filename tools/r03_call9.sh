#!/bin/bash
# round 3, call 9: mirrored (bottom-up) odd row blocks: parity with the dev build, then A/B timing (plain kernel and unit step)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call9
rm -rf $out && mkdir -p $out
cd $root
export NUS_LIB_PATH=$root/tools/_ablate/lib_flip.so
timeout -k 10 900 python -m pytest tests/test_unit_step.py tests/test_gpu_parity.py -x -q -m gpu -k "unit or lanczos or config2 or bench_launch or fused or opaque or bgra or resize or bicubic" > $out/flip_tests.txt 2>&1; rc=$?; echo "flip tests rc=$rc"; tail -6 $out/flip_tests.txt
unset NUS_LIB_PATH
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/lz_variants.py --rounds 5 cur=nu_scaler_amd/lib/libnuscaler_hip.so flip=tools/_ablate/lib_flip.so > $out/flip_plain_ab.txt 2>&1; echo "rc=$?"; grep -v amdgpu $out/flip_plain_ab.txt
for v in product flip product flip; do
  echo "== unit step: $v"
  if [ $v = flip ]; then export NUS_LIB_PATH=$root/tools/_ablate/lib_flip.so; else unset NUS_LIB_PATH; fi
  timeout -k 10 300 python3 tools/unit_bench.py --rounds 3 2>&1 | grep median
done > $out/flip_unit_ab.txt 2>&1; unset NUS_LIB_PATH; cat $out/flip_unit_ab.txt
