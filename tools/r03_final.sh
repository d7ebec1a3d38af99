#!/bin/bash
# round 3, last call: the whole GPU suite, the stress sweeps, smoke, then the round's evidence
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
export NUS_EXPECT_GPU=1
mkdir -p gpurun_out/r03_final
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03_final/gpu_tests.txt 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 gpurun_out/r03_final/gpu_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 600 python3 tools/stress_unit_step.py 120 777 2>&1 | tail -2
timeout -k 10 600 python3 tools/stress_fixed_factors.py 2>&1 | tail -2
timeout -k 10 600 python3 tools/stress_flow.py 2>&1 | tail -2
timeout -k 10 900 python3 tools/stress_batch_consistency.py 2>&1 | tail -1
bash tools/round_evidence.sh
