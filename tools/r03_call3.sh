#!/bin/bash
# round 3, call 3: pipelined upscale_batch (tests + host path numbers) and the bench line on the unit schedule
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call3
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch or concurrent or errors_and_reinit or interp_fixture" > $out/batch_tests.txt 2>&1; rc=$?; echo "batch tests rc=$rc"; tail -5 $out/batch_tests.txt
[ $rc -eq 0 ] || exit 1
for t in 3 5 7; do echo "== NUS_COPY_THREADS=$t"; NUS_COPY_THREADS=$t timeout -k 10 200 python3 tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids; done > $out/host_path.txt 2>&1; cat $out/host_path.txt
timeout -k 10 600 python3 bench.py > $out/bench_n1.log 2>&1; echo "bench rc=$?"; tail -1 $out/bench_n1.log > $out/bench_n1.json; cut -c1-300 $out/bench_n1.json; tail -5 $out/bench_n1.log | cut -c1-300 | grep -v "^{" 
