#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X upscale + interpolation hot path.

Workload (BASELINE.json `metric`, configs[4] sized for one GPU): a synthetic 1080p
RGBA8 stream resident in HBM; one *unit* = one source frame k: warp+blend (k, k+1) at
t = 0.5 into an in-between frame, then Lanczos-3 x2 of frame k and of the in-between
frame to 3840x2160.  One *step* = one pass over the per-GPU batch of units.
Metric: Mpixels/s, input + output pixels of every kernel counted once each
(26.9568 Mpix per unit).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL; frames are sharded
   contiguously across ranks -- weak scaling, no data-path collective; the only
   collective is the one-off broadcast of the filter tables.)

Rank 0 prints ONE JSON line.  `roofline` is measured live for the dominant kernel
(k_lanczos3_x2) with hipEvent pairs on the launch stream inside the timed region;
`cpu_baseline` times the CPU oracle (a port of the reference's CPU algorithm -- the
Rust reference cannot be built here) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", type=int, default=300, help="source frames per GPU per step (the 300-frame stream)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pattern", choices=["gradient", "noise"], default="gradient")
    ap.add_argument("--lanczos-mode", choices=["fma", "exact"], default="fma")
    ap.add_argument("--cpu-baseline-units", type=int, default=-1,
                    help="units timed on the CPU oracle; -1 = sized for ~15 s of single-thread work, 0 disables")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc traffic passes (N=1 only)")
    ap.add_argument("--no-profile", action="store_true", help="skip the in-loop hipEvent pairs")
    ap.add_argument("--fused", action="store_true",
                    help="blend inside the second upscale's row loads (in-between frame never written to HBM); "
                         "same output frames, reported separately from the default 3-stage step")
    ap.add_argument("--overlap", action="store_true",
                    help="blend on a second stream, concurrent with the upscale of the real frames (measured: no gain, "
                         "the Lanczos kernel is SIMD-time bound and slows by what the blend takes)")
    ap.add_argument("--motion", action="store_true",
                    help="also time the motion-compensated step (per-pair pyramid + Horn-Schunck flow feeding the warp); "
                         "informational leg config.motion_variant, never `value`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--force-device", type=int, default=-1,
                    help="rehearsal only: put every rank on this GPU (with --backend gloo on a 1-GPU box)")
    return ap.parse_args()


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file ships with the repo)."""
    fallback = "Mpixels/sec (in+out) at 1080p\u21924K \u00d72 upscale + interp, 1/2/4/8 GPU"
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f).get("metric", fallback)
    except (OSError, ValueError):
        return fallback


def cpu_baseline(args, unit_pixels):
    """Time the CPU oracle on a bounded sample of the same workload (single thread, as the
    reference's BasicUpscaler runs; plus an all-cores OpenMP figure for context)."""
    import oracle

    oracle.build()
    w, h = args.width, args.height

    def run(n, threads):
        frames = [oracle.gen_gradient(w, h, k) for k in range(n + 1)]
        t0 = time.perf_counter()
        for k in range(n):
            mid = oracle.warp_blend(frames[k], frames[k + 1], None, 0.5, threads=threads)
            oracle.lanczos3(frames[k], 2 * w, 2 * h, threads=threads)
            oracle.lanczos3(mid, 2 * w, 2 * h, threads=threads)
        return time.perf_counter() - t0

    n = args.cpu_baseline_units
    if n < 0:  # size the sample for ~15 s of single-thread work
        t_unit = run(1, 1)
        n = max(4, min(64, int(round(15.0 / max(t_unit, 1e-3)))))
    t1 = run(n, 1)
    cores = oracle.max_threads()
    tn = run(n, 0) if cores > 1 else t1
    return {
        "value": round(n * unit_pixels / t1 / 1e6, 3),
        "unit": "Mpix/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} units of the same stream ({w}x{h}: zero-flow blend + 2x Lanczos-3 x2), oracle/nus_oracle.c "
                  f"gcc -O2 -ffp-contract=off, {t1:.1f} s single thread",
        "all_cores": {"value": round(n * unit_pixels / tn / 1e6, 3), "cores": cores, "seconds": round(tn, 2)},
    }


def measure_traffic(frames_per_launch):
    """HBM bytes per launch of the dominant kernel from the L2's memory-side counters, in two
    separate rocprofv3 --pmc passes over a kernel-only child process (never combined with
    tracing).  gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE
    reports half the bytes of a wide coalesced read (x2); WRITE_SIZE is exact; both in KiB.
    Returns (bytes_per_launch_scaled_to_frames_per_launch, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    n_child = 64
    vals = {}
    tmp = tempfile.mkdtemp(prefix="nus_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "tools", "lanczos_only.py"), str(n_child), "2"]
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=tmp)
            if res.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} failed (rc {res.returncode})"
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_lanczos3_x2<" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        rows.append(float(r["Counter_Value"]))
            if not rows:
                return None, f"no {counter} rows for k_lanczos3_x2"
            vals[counter] = sum(rows) / len(rows)
    except Exception as e:  # timeouts, missing files: traffic stays null
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per_frame = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 / n_child
    detail = {"FETCH_SIZE_KiB_per_frame": round(vals["FETCH_SIZE"] / n_child, 1),
              "WRITE_SIZE_KiB_per_frame": round(vals["WRITE_SIZE"] / n_child, 1),
              "fetch_correction": 2.0, "child_frames_per_launch": n_child}
    return int(per_frame * frames_per_launch), detail


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import nu_scaler_amd as nsc
    from nu_scaler_amd import synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    nccl = args.backend == "nccl"
    comm_dev = dev if nccl else torch.device("cpu")  # gloo rehearsal: collectives on host tensors
    if world > 1:
        if nccl:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    w, h = args.width, args.height
    pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5, device=local_rank, lanczos_mode=args.lanczos_mode)
    # shared LUTs: rank 0's tables to everyone (RCCL over xGMI), so all GPUs use identical weights
    lut_bytes = nsc.broadcast_tables(pipe.upscaler, 0, comm_dev)

    # this rank's contiguous shard of the global stream, plus the overlap frame
    n_units = args.units
    start, count = nsc.shard_frames(n_units * world, world, rank)
    assert count == n_units
    gen = syn.gradient_stream_torch if args.pattern == "gradient" else None
    frames = torch.empty((count + 1, h, w, 4), dtype=torch.uint8, device=dev)
    for c0 in range(0, count + 1, 16):  # generate in chunks: int64 temporaries are 8x a frame
        c1 = min(c0 + 16, count + 1)
        if gen is not None:
            frames[c0:c1] = gen(c1 - c0, w, h, dev, first=start + c0)
        else:
            frames[c0:c1] = syn.noise_stream_torch(c1 - c0, w, h, dev, seed=0x5EED + start + c0)
    mid, up_real, up_mid = pipe.alloc(count, dev)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            if nccl:
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    def do_step():
        if args.fused:
            pipe.step_fused(frames, up_real, up_mid, stream)
        elif args.overlap:
            pipe.step_overlapped(frames, mid, up_real, up_mid)
        else:
            pipe.step(frames, mid, up_real, up_mid, stream)

    for _ in range(args.warmup):
        do_step()
    torch.cuda.synchronize()
    profile = not args.no_profile
    pipe.upscaler.set_profiling(profile)
    pipe.upscaler.profile_collect()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        do_step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    launches, kernel_ms = pipe.upscaler.profile_collect() if profile else (0, 0.0)
    pipe.upscaler.set_profiling(False)

    # Informational second leg (never `value`): the same output frames with the blend fused into the
    # second upscale's row loads, so the 1080p in-between frame is never written to HBM.
    fused_ms = None
    if not args.fused and not args.overlap:
        for _ in range(2):
            pipe.step_fused(frames, up_real, up_mid, stream)
        torch.cuda.synchronize()
        tf = time.perf_counter()
        for _ in range(max(3, args.steps // 4)):
            pipe.step_fused(frames, up_real, up_mid, stream)
        torch.cuda.synchronize()
        fused_ms = (time.perf_counter() - tf) / max(3, args.steps // 4) * 1e3

    # Informational (never `value`): the on-box ceiling for these very bytes -- k_nearest_x2 reads the same
    # 1080p frames and writes the same 4K frames with no arithmetic -- so roofline.frac can be read next
    # to what the memory system of this box actually sustains, not only next to the 8 TB/s spec figure.
    copy_ms = None
    if rank == 0 and profile:
        nn = nsc.PyWgpuUpscaler("quality", "nearest", device=local_rank)
        nn.initialize(w, h, 2 * w, 2 * h)
        nn.set_profiling(True)
        for _ in range(2):
            nn.upscale_device(frames.data_ptr(), up_real.data_ptr(), count, stream)
        torch.cuda.synchronize()
        nn.profile_collect()
        for _ in range(5):
            nn.upscale_device(frames.data_ptr(), up_real.data_ptr(), count, stream)
        torch.cuda.synchronize()
        nl, nms = nn.profile_collect()
        copy_ms = nms / max(nl, 1)
        del nn

    t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # cheap sanity check outside the timed region: the stream is a 1 px/frame shift, so
    # the outputs must differ between units and be fully written (alpha stays 255)
    assert int(up_real[0, ..., 3].min()) == 255 and int(up_mid[count - 1, ..., 3].min()) == 255

    # (after the check above: with a real flow the bilinear samples' truncation can take alpha to 254)
    motion_ms = None
    if args.motion and rank == 0:
        flows = torch.empty((count, h, w, 2), dtype=torch.float32, device=dev)
        pipe.step_motion(frames, flows, mid, up_real, up_mid, stream)
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for _ in range(2):
            pipe.step_motion(frames, flows, mid, up_real, up_mid, stream)
        torch.cuda.synchronize()
        motion_ms = (time.perf_counter() - tm) / 2 * 1e3
        del flows


    if rank == 0:
        total_units = n_units * world * args.steps
        value = total_units * pipe.unit_pixels / elapsed / 1e6
        up_bytes = (w * h + 4 * w * h) * 4  # algorithmic bytes of one upscaled frame
        roofline = None
        if launches:
            per_launch_s = kernel_ms / 1e3 / launches
            achieved = up_bytes * count / per_launch_s / 1e9
            roofline = {
                "bound": "hbm", "kernel": "k_lanczos3_x2", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "bytes_per_launch": up_bytes * count, "frames_per_launch": count, "launches": launches,
                "avg_launch_ms": round(kernel_ms / launches, 4),
                "note": "achieved = algorithmic bytes (8 294 400 R + 33 177 600 W per frame) / hipEvent time of the "
                        "main kernel on its launch stream inside the timed region; traffic = (2*FETCH_SIZE + "
                        "WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes, scaled to frames_per_launch",
            }
            if copy_ms:
                ceiling = up_bytes * count / (copy_ms / 1e3) / 1e9
                roofline["copy_ceiling"] = {
                    "GBps": round(ceiling, 1), "frac_of_ceiling": round(achieved / ceiling, 4),
                    "how": "k_nearest_x2 over the same frames (same bytes in and out, no arithmetic), hipEvent time"}
        out = {
            "metric": baseline_metric(),
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 frames, f32 taps",
            "data": "synthetic",
            "config": {
                "workload": f"{n_units}-frame {w}x{h} RGBA8 stream per GPU, HBM-resident: zero-flow warp+blend t=0.5 "
                            f"(1 in-between frame per source frame) + Lanczos-3 x2 of real and in-between frame to "
                            f"{2 * w}x{2 * h}",
                "units_per_step_per_gpu": n_units, "pixels_per_unit": pipe.unit_pixels,
                "algorithmic_bytes_per_unit": pipe.unit_bytes, "pattern": args.pattern,
                "lanczos_mode": args.lanczos_mode, "kernel_variant": pipe.upscaler.kernel_variant,
                "schedule": "fused: upscale(real) + upscale(blend(A,B)) with the blend in the row loads, 2 launches"
                            if args.fused else
                            "blend on a second HIP stream, concurrent with the upscale of the real frames"
                            if args.overlap else "3 stages back to back on one stream",
                "sharding": f"frame-parallel, contiguous shards, {world} rank(s), LUT broadcast {lut_bytes} B over "
                            f"{'RCCL' if nccl else args.backend}",
                "fused_variant": None if fused_ms is None else {
                    "what": "same 4K outputs, blend fused into the second upscale (in-between frame not materialised); "
                            "informational, measured after the timed region on this rank only",
                    "ms_per_step": round(fused_ms, 4),
                    "Mpix_per_s_per_gpu_same_unit_pixels": round(n_units * pipe.unit_pixels / fused_ms / 1e3, 1)},
                "motion_variant": None if motion_ms is None else {
                    "what": "same step with a dense flow per pair (3-level pyramid, 50 + 10 + 10 Horn-Schunck steps) feeding "
                            "the warp instead of zero flow; informational, this rank only",
                    "ms_per_step": round(motion_ms, 3),
                    "units_per_s_per_gpu": round(n_units / motion_ms * 1e3, 1)},
                "frames_per_sec_per_gpu_4k_out": round(2 * n_units * args.steps / elapsed, 1),
                "algorithmic_GBps": round(total_units * pipe.unit_bytes / elapsed / 1e9, 1),
            },
            "roofline": roofline,
        }
        if world == 1 and roofline is not None and not args.no_pmc:
            traffic, detail = measure_traffic(count)
            roofline["traffic"] = traffic
            roofline["traffic_detail"] = detail
        if world == 1 and args.cpu_baseline_units != 0:
            out["cpu_baseline"] = cpu_baseline(args, pipe.unit_pixels)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
