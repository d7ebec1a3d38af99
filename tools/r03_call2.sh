#!/bin/bash
# round 3, call 2: the one-launch unit step: parity tests, then timing against the three-stage and fused schedules
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call2
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests/test_unit_step.py -x -q -m gpu > $out/unit_tests.txt 2>&1; rc=$?; echo "unit tests rc=$rc"; tail -15 $out/unit_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/unit_bench.py > $out/unit_bench.txt 2>&1; echo "unit bench rc=$?"; cat $out/unit_bench.txt | grep -v amdgpu.ids
timeout -k 10 300 python3 tools/unit_bench.py --no-mid --rounds 3 > $out/unit_bench_nomid.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $out/unit_bench_nomid.txt
timeout -k 10 300 python3 tools/unit_bench.py --th 18 --rounds 3 > $out/unit_bench_th18.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $out/unit_bench_th18.txt
