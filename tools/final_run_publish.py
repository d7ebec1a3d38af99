#!/usr/bin/env python
"""Copies a finished evidence run (gpurun_out/evidence, tools/round_evidence.sh a + b) into profiles/rNN_final_run/, writes its README.txt (the documents quote the round's boxes by hand).  final_run_publish.py <round number>   (dev tool)"""
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
src, dst = os.path.join(root, "gpurun_out", "evidence"), os.path.join(root, "profiles", f"r{rnd}_final_run")
os.makedirs(dst, exist_ok=True)
keep = ("bench_driver_command_steps20.json bench_n1.json bench_n1_kernel_stats.csv bench_under_rocprof.json bench_three_stage_kernel_stats.csv "
        "bench_three_stage_under_rocprof.json unit_step_kernel_timeline.txt unit_step_schedules.txt quick_bench_kernels.txt warp_kernel.txt host_path.txt "
        "general_scale_sweep.txt pq_factors_vs_any_scale_kernel.txt nearest_bilinear_pq_ratios.txt rcas_rows.txt flow_kernels.txt flow_stream_kernels.txt "
        "flow_stream_kernels_fast_mode.txt flow_stream_exact_vs_fast_vs_shifting_fast.txt flow_front_end_kernel_stats.csv motion_step_pipelined.txt "
        "edge_stream_ab.txt bench_rehearsal_n2_gloo_one_gpu.json bench_rehearsal_n4_gloo_one_gpu.json").split()
for f in keep:
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for name, to in (("r06_gpu_tests_final.txt", "gpu_tests.txt"), ("r06_smoke.txt", "smoke.txt")):
    p = os.path.join(root, "gpurun_out", name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, to))
readme = subprocess.run([sys.executable, os.path.join(root, "tools", "final_run_readme.py"), src, rnd], capture_output=True, text=True, check=True).stdout
open(os.path.join(dst, "README.txt"), "w").write(readme)
print(readme[:900])
