#!/bin/bash
# Collects the round's judged evidence on the GPU box into gpurun_out/evidence/ (copy what you keep into profiles/):
# the bench line, rocprofv3 --stats of the same command, the schedules side by side, the per-kernel quick bench (both patterns),
# the warp kernel, the general-factor sweep, the flow front end's kernel breakdown, the 2-rank rehearsal.
# usage: tools/round_evidence.sh [a|b|all]   -- two halves, so that each fits one gpurun call (20 minutes): a = the bench line, its profiles,
# the schedules, the per-kernel quick bench, warp, host path; b = the factor sweeps, the flow front end, the motion step, the rehearsals
part=${1:-all}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/evidence
if [ $part != b ]; then rm -rf $out; fi
mkdir -p $out
cd $root
if [ $part != b ]; then
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command.log 2>&1 && grep "^{" $out/bench_driver_command.log | tail -1 > $out/bench_driver_command_steps20.json
echo "driver command done: $(cut -c1-200 $out/bench_driver_command_steps20.json)"
python3 bench.py > $out/bench_n1.log 2>&1 && grep "^{" $out/bench_n1.log | tail -1 > $out/bench_n1.json
echo "bench done: $(cut -c1-160 $out/bench_n1.json)"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -o b -- python3 $root/bench.py --no-pmc --no-cpu-baseline --no-extras --no-check --sustained-seconds 0 --steps 450 --warmup 30 > $out/bench_prof.log 2>&1)
grep "^{" $out/bench_prof.log | tail -1 > $out/bench_under_rocprof.json
cp $(find $out/bench_prof -name "*kernel_stats.csv" | head -1) $out/bench_n1_kernel_stats.csv 2>/dev/null
echo "bench profile done"; head -6 $out/bench_n1_kernel_stats.csv
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench3_prof -o b -- python3 $root/bench.py --schedule three-stage --no-pmc --no-cpu-baseline --no-extras --no-check --sustained-seconds 0 --steps 200 --warmup 30 > $out/bench3_prof.log 2>&1)
grep "^{" $out/bench3_prof.log | tail -1 > $out/bench_three_stage_under_rocprof.json
cp $(find $out/bench3_prof -name "*kernel_stats.csv" | head -1) $out/bench_three_stage_kernel_stats.csv 2>/dev/null
echo "three-stage profile done"
python3 tools/unit_bench.py 2>&1 | grep -v amdgpu.ids > $out/unit_step_schedules.txt
echo "schedules done"
for pat in gradient noise; do python3 tools/quick_bench.py --frames 300 --reps 5 --pattern $pat 2>&1 | grep -v amdgpu.ids; done > $out/quick_bench_kernels.txt
echo "quick bench done"
python3 tools/warp_bench.py 2>&1 | grep -v amdgpu.ids > $out/warp_kernel.txt
python3 tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids > $out/host_path.txt
fi
if [ $part != a ]; then
python3 tools/general_sweep.py > $out/general_scale_sweep.txt 2>&1
python3 tools/pq_bench.py 128 2>&1 | grep -v amdgpu.ids > $out/pq_factors_vs_any_scale_kernel.txt
bash tools/ratio_bench.sh 2>&1 | grep -v amdgpu.ids > $out/nearest_bilinear_pq_ratios.txt
python3 tools/rcas_bench.py 60 2>&1 | grep -v amdgpu.ids > $out/rcas_rows.txt
echo "sweep done"
bash tools/flow_prof.sh > $out/flow_kernels.txt 2>&1
cp $(find $root/gpurun_out/flowprof -name "*kernel_stats.csv" | head -1) $out/flow_front_end_kernel_stats.csv 2>/dev/null
python3 tools/flow_bench.py 2>&1 | grep "flow estimate" >> $out/flow_kernels.txt
bash tools/flow_stream_prof.sh 101 3 > $out/flow_stream_kernels.txt 2>&1
bash tools/flow_stream_prof.sh 101 9 > $out/flow_stream_kernels_fast_mode.txt 2>&1
python3 tools/flow_stream_bench.py 101 3 9 19 2>&1 | grep "flow stream" > $out/flow_stream_exact_vs_fast_vs_shifting_fast.txt
python3 tools/motion_bench.py 300 100 2>&1 | grep "motion step" > $out/motion_step_pipelined.txt
python3 tools/motion_default.py 2>&1 | grep "motion step" >> $out/motion_step_pipelined.txt
echo "flow done"
python3 tools/edge_stream_ab.py 2>&1 | grep -v amdgpu > $out/edge_stream_ab.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out/edge_prof -o e -- python3 $root/tools/unit_only.py 300 3 > $out/edge_prof.log 2>&1)
python3 - "$(find $out/edge_prof -name "*kernel_trace.csv" | head -1)" > $out/unit_step_kernel_timeline.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_lanczos3_x2" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
print("# rocprofv3 --kernel-trace of tools/unit_only.py 300 3: the one-launch step's three kernels, start / end in us from the first start")
print("# (the edge-column passes run on the upscaler's second stream BESIDE the unit kernel: their durations overlap it and must not be added to it)")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s_, e_ = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = "edges" if "_edges" in r["Kernel_Name"] else "unit "
    print(f"{name}  queue {r.get('Queue_Id', '?'):>3s}  start {s_:10.1f}  end {e_:10.1f}  duration {e_ - s_:8.1f} us")
PY
rm -rf $out/edge_prof
python3 bench.py --gpus 2 --backend gloo --force-device 0 --steps 30 --warmup 3 --units 100 --sustained-seconds 2 > $out/bench_n2_gloo.log 2>&1; grep "^{" $out/bench_n2_gloo.log | tail -1 > $out/bench_rehearsal_n2_gloo_one_gpu.json
echo "rehearsal done: $(cut -c1-120 $out/bench_rehearsal_n2_gloo_one_gpu.json)"
timeout -k 10 400 python3 bench.py --gpus 4 --backend gloo --force-device 0 --steps 20 --warmup 3 --units 60 --sustained-seconds 2 --host-fed-seconds 1 > $out/bench_n4_gloo.log 2>&1; grep "^{" $out/bench_n4_gloo.log | tail -1 > $out/bench_rehearsal_n4_gloo_one_gpu.json
echo "4-rank rehearsal done: $(cut -c1-120 $out/bench_rehearsal_n4_gloo_one_gpu.json)"
fi
rm -rf $out/bench_prof $out/bench3_prof $root/gpurun_out/flowsprof $root/gpurun_out/flowprof
ls -la $out
