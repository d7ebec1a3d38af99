#!/bin/bash
# round 3, first GPU call: what the current step sustains (bench line with the sustained leg), RCCL one-rank test,
# PCIe ceilings and the host path against copy threads
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03_call1
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 300 python -m pytest tests/test_rccl_one_rank.py -x -q -m gpu > $out/rccl_test.txt 2>&1; echo "rccl test rc=$?"; tail -3 $out/rccl_test.txt
timeout -k 10 500 python3 bench.py > $out/bench_n1.log 2>&1; echo "bench rc=$?"; tail -1 $out/bench_n1.log > $out/bench_n1.json; cut -c1-400 $out/bench_n1.json
timeout -k 10 120 python3 tools/pcie_probe.py > $out/pcie_probe.txt 2>&1; cat $out/pcie_probe.txt
for t in 3 7; do echo "== NUS_COPY_THREADS=$t"; NUS_COPY_THREADS=$t timeout -k 10 200 python3 tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids; done > $out/host_path.txt 2>&1; cat $out/host_path.txt
