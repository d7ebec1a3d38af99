#!/usr/bin/env python
"""Writes the README.txt of a round's final evidence run from the files tools/round_evidence.sh collected:
final_run_readme.py <evidence dir> <round number>  > profiles/rNN_final_run/README.txt   (dev tool)"""
import csv
import json
import os
import sys

d, rnd = sys.argv[1], sys.argv[2]


def line(name):
    p = os.path.join(d, name)
    return json.load(open(p)) if os.path.exists(p) and os.path.getsize(p) else None


def kernel_avg_ms(name, needle):
    p = os.path.join(d, name)
    if not os.path.exists(p):
        return None
    for r in csv.reader(open(p)):
        if r and needle in r[0]:
            return int(r[1]), float(r[3]) / 1e6
    return None


drv, dflt, prof, three = (line(n) for n in ("bench_driver_command_steps20.json", "bench_n1.json", "bench_under_rocprof.json",
                                             "bench_three_stage_under_rocprof.json"))
print(f"profiles/r{rnd}_final_run/ -- collected by tools/round_evidence.sh (two gpurun calls: a, b) after the last code change of round {int(rnd)}.")
print("The headline kernel (k_lanczos3_x2<FMA, blend-on-load, UNIT>) was not changed in this round; boxes of the pool differ by 7-15 % on one binary.\n")
for what, x in (("driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5)", drv), ("450-step default (python3 bench.py)", dflt),
                ("under rocprofv3 --kernel-trace --stats (450 steps)", prof)):
    if x:
        r = x["roofline"]
        print(f"{what}: {x['value']:.0f} Mpix/s, {x['ms_per_step']:.4f} ms per step, roofline.frac {r['frac']}, moved_frac {r.get('moved_frac')}, "
              f"bracket {r.get('avg_launch_ms')} ms")
u = kernel_avg_ms("bench_n1_kernel_stats.csv", "k_lanczos3_x2<false, 1, true, false>")
if u and prof:
    print(f"\nbench_n1_kernel_stats.csv: k_lanczos3_x2<false, 1, true, false> {u[0]} calls, average {u[1]:.4f} ms -- the hipEvent bracket of that run: "
          f"{prof['roofline'].get('avg_launch_ms')} ms.  The two edge-column kernels run on the upscaler's second stream BESIDE the unit kernel: their reported "
          "durations overlap it and must NOT be added to it (unit_step_kernel_timeline.txt).")
p3 = kernel_avg_ms("bench_three_stage_kernel_stats.csv", "k_lanczos3_x2<false, 0, false, false>")
qb = None
qp = os.path.join(d, "quick_bench_kernels.txt")
if os.path.exists(qp):
    for ln in open(qp):
        if ln.startswith("lanczos3") and "'fma'" in ln and qb is None:
            qb = float(ln.split("us/frame")[0].split()[-1])
if p3 and qb:
    us = p3[1] * 1e3 / 300
    print(f"\nquick_bench.py against rocprofv3 in the same call (VERDICT r05 item 5): k_lanczos3_x2 FMA on the gradient stream {qb:.2f} us per frame in "
          f"quick_bench_kernels.txt, {us:.2f} in bench_three_stage_kernel_stats.csv ({p3[0]} calls): {100 * (qb / us - 1):+.1f} %.")
if dflt:
    c, r = dflt["config"], dflt["roofline"]
    m = c.get("motion_variant") or {}
    c3 = r.get("config3") or {}
    print(f"\nbench_n1.json: config3 (the plain Lanczos stream, 3 s): gradient {c3.get('gradient', {}).get('us_per_frame')} us per frame = "
          f"{c3.get('gradient', {}).get('frac')}, noise {c3.get('noise', {}).get('us_per_frame')} = {c3.get('noise', {}).get('frac')}; "
          f"motion_variant ({m.get('configuration')}): {m.get('ms_per_step')} ms per 300 units (f32 hand-off {m.get('pipelined_f32_handoff_ms_per_step')}, "
          f"stage after stage {m.get('stage_by_stage_ms_per_step')}, exact flow {m.get('exact_flow_ms_per_step')}); cpu_baseline "
          f"{dflt['cpu_baseline']['value']} Mpix/s on {dflt['cpu_baseline']['cores']} core(s); copy ceiling: stream copy "
          f"{r['copy_ceiling'].get('stream_copy_float4_GBps')} GB/s, 1 R : 4 W {r['copy_ceiling'].get('one_read_four_writes_GBps')}, k_nearest_x2 "
          f"{r['copy_ceiling'].get('k_nearest_x2_GBps')}.")
print("""
Files: bench_driver_command_steps20.json, bench_n1.json (the lines); bench_n1_kernel_stats.csv / bench_under_rocprof.json and
bench_three_stage_* (rocprofv3 --kernel-trace --stats of the same step, both schedules); unit_step_kernel_timeline.txt; unit_step_schedules.txt;
quick_bench_kernels.txt (per kernel, both patterns; since round 6: a second of warm launches per case, collectors reset, median of 5);
warp_kernel.txt; host_path.txt; general_scale_sweep.txt; pq_factors_vs_any_scale_kernel.txt (since round 6 with the 4-tap and the 6-tap
form of the P/Q kernel side by side for bicubic); nearest_bilinear_pq_ratios.txt; rcas_rows.txt; flow_kernels.txt, flow_stream_kernels*.txt,
flow_stream_exact_vs_fast_vs_shifting_fast.txt, flow_front_end_kernel_stats.csv; motion_step_pipelined.txt (its last two lines: the default
configuration, Rg16Float / f32 hand-off); edge_stream_ab.txt; bench_rehearsal_n2_gloo_one_gpu.json / _n4_ (two / four gloo ranks on the one GPU).
Round 6's other evidence sits beside this directory: r06_pageable_d2h_probe.txt, r06_flow_level0_luminance_in_jacobi_ablation.txt,
r06_flow_nontemporal_ab.txt, r06_flow_level0_half_between_launches.txt, r06_nearest_table_staged_rows_ab.txt, r06_x2_row_store_without_lds_ab.txt.""")
