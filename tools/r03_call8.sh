#!/bin/bash
# round 3, call 8: host pipeline on three shared streams: tests, host path numbers, bench host legs
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call8
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_imagefile.py -x -q -m gpu -k "batch or concurrent or errors_and_reinit or upscale or c_program or cli" > $out/batch_tests.txt 2>&1; rc=$?; echo "batch tests rc=$rc"; tail -4 $out/batch_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python3 tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids > $out/host_path.txt; cat $out/host_path.txt
timeout -k 10 400 python3 tools/host_path_state_probe.py 2>&1 | grep -v amdgpu > $out/host_path_state.txt; cat $out/host_path_state.txt
timeout -k 10 600 python3 bench.py --no-pmc --no-cpu-baseline --sustained-seconds 0 --steps 100 > $out/bench_short.log 2>&1; echo "bench rc=$?"; tail -1 $out/bench_short.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:v for k,v in d['config']['host_path'].items() if k!='what'})); print(d['config']['host_fed'])"
