#!/bin/bash
# Collects the round's judged evidence on the GPU box into gpurun_out/evidence/ (copy what you keep into profiles/):
# the bench line, rocprofv3 --stats of the same command, the per-kernel quick bench (both patterns), the general-factor
# sweep and the flow front end's kernel breakdown.
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/evidence
rm -rf $out && mkdir -p $out
cd $root
python3 bench.py > $out/bench_n1.log 2>&1 && tail -1 $out/bench_n1.log > $out/bench_n1.json
echo "bench done: $(cut -c1-120 $out/bench_n1.json)"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -o b -- python3 $root/bench.py --no-pmc --no-cpu-baseline --no-extras --no-check --steps 20 --warmup 3 > $out/bench_prof.log 2>&1)
grep "^{" $out/bench_prof.log | tail -1 > $out/bench_under_rocprof.json
cp $(find $out/bench_prof -name "*kernel_stats.csv" | head -1) $out/bench_n1_kernel_stats.csv 2>/dev/null
echo "bench profile done"
for pat in gradient noise; do python3 tools/quick_bench.py --frames 300 --reps 5 --pattern $pat 2>&1 | grep -v amdgpu.ids; done > $out/quick_bench_kernels.txt
echo "quick bench done"
python3 tools/general_sweep.py > $out/general_scale_sweep.txt 2>&1
echo "sweep done"
bash tools/flow_prof.sh > $out/flow_kernels.txt 2>&1
cp $(find $root/gpurun_out/flowprof -name "*kernel_stats.csv" | head -1) $out/flow_front_end_kernel_stats.csv 2>/dev/null
python3 tools/flow_bench.py 2>&1 | grep "flow estimate" >> $out/flow_kernels.txt
bash tools/flow_stream_prof.sh 65 > $out/flow_stream_kernels.txt 2>&1
echo "flow done"
rm -rf $out/bench_prof
ls -la $out
