#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X upscale + interpolation hot path.

Workload (BASELINE.json `metric`, configs[4] sized for one GPU): a synthetic 1080p
RGBA8 stream resident in HBM; one *unit* = one source frame k: warp+blend (k, k+1) at
t = 0.5 into an in-between frame, then Lanczos-3 x2 of frame k and of the in-between
frame to 3840x2160.  One *step* = one pass over the per-GPU batch of units.
Metric: Mpixels/s, input + output pixels of every kernel counted once each
(26.9568 Mpix per unit).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL; frames are sharded contiguously across ranks (weak
scaling, no data-path collective; the only collective is the one-off broadcast of the
filter tables).  Launched by `python -m torch.distributed.run ... bench.py --gpus N`
the ranks find WORLD_SIZE in the environment; started directly (`python bench.py --gpus 4`)
this process only starts that launcher as a CHILD, relays rank 0's JSON line and the exit
code, and never touches the GPU itself.

The timed step is the ONE-LAUNCH schedule (`--schedule unit`, nus_upscaler_upscale_unit_device): every (frame, row block,
strip) is walked by two waves of k_lanczos3_x2 side by side -- one up-scaling frame k, one blending frames k and k+1 on load,
storing the in-between rows and up-scaling them -- so all three outputs of a unit (the in-between frame and both 4K frames) come
from one kernel; `--schedule three-stage` is the blend / upscale / upscale sequence of rounds 1-2 (same bytes out).

Rank 0 prints ONE JSON line.  Inside it:
  roofline       the dominant launch (the unit kernel + its two edge-column passes), hipEvent pairs on the launch stream
                 inside the timed region, algorithmic bytes of SURVEY.md section 8(d) against the 8 TB/s HBM peak; next to it
                 `upscale_kernel_alone`: k_lanczos3_x2 by itself on the same frames (41 472 000 B per frame);
  cpu_baseline   the CPU oracle (a port of the reference's CPU algorithm -- the Rust reference
                 cannot be built here), per algorithm, median of 10 after 2 warm-ups;
  config.timed_output_check   frames of the TIMED output buffers compared with the oracle;
  config.noise_variant        the same step on 4-channel noise (the kernel's other code path);
  config.host_path            host bytes in -> host bytes out through the trait-shaped entry
                              points (PCIe inclusive; never `value`).
  config.placement / per_rank / per_rank_check   what happened on EVERY rank: the NUMA node and CPUs it was bound to before
                              its first HIP call, its time, its bracket, its clocks, and two frames of ITS shard (all three
                              output buffers) against the oracle -- gathered, so rank 0's line can show an outlier.
  roofline.copy_ceiling       what hipMemcpyDtoDAsync, a 16-B-per-lane stream copy and a 1 R : 4 W stream get on this box
                              (SURVEY.md 8d's on-box ceiling), and `roofline.moved`: the bytes the kernel really moves
                              (`traffic`) against them;  roofline.config3: BASELINE config 3, the plain Lanczos stream,
                              sustained for seconds by itself.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("LIBC_FATAL_STDERR_", "1")  # a fatal glibc message belongs in the run's stderr, not on a terminal nobody reads

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def _install_crash_reports():
    """faulthandler (Python frames of every thread); the library's own fatal-signal report follows in _install_fatal_trace
    (nus_install_fatal_trace: native backtrace of the raising thread, the host ranges the library holds, /proc/self/maps):
    a rank that dies says which library aborted and where the address in the runtime's last words lies."""
    import faulthandler

    faulthandler.enable(file=sys.__stderr__, all_threads=True)


def _install_fatal_trace():
    """The library's half of the crash report.  Called AFTER the ShardedStream exists: loading libnuscaler_hip.so pulls in a HIP
    runtime, and torch -- which must be the first to load one -- is imported by the ShardedStream only after the rank has bound
    itself to its GPU's NUMA node and sized OMP_NUM_THREADS (libgomp reads it when it is loaded)."""
    import nu_scaler_amd

    nu_scaler_amd.install_fatal_trace(2)


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~3 s of timed region on one MI355X (6-7 ms per 300-unit step), so `value` is what the step
    # sustains at the clocks the board settles to under it, not what a cold GPU does for a tenth of a second
    ap.add_argument("--steps", type=int, default=450)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--units", type=int, default=300, help="source frames per GPU per step (the 300-frame stream)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pattern", choices=["gradient", "noise"], default="gradient")
    ap.add_argument("--lanczos-mode", choices=["fma", "exact"], default="fma")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle timing (N=1 only leg)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc traffic passes (N=1 only)")
    ap.add_argument("--no-profile", action="store_true", help="skip the in-loop hipEvent pairs")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational legs (fused / noise / motion / host path / copy ceiling)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle check of the timed output buffers")
    ap.add_argument("--sustained-seconds", type=float, default=6.0,
                    help="length of the sustained leg after the timed region (0 = skip; every rank at the same time)")
    ap.add_argument("--host-fed-seconds", type=float, default=1.5,
                    help="length of the host-fed leg (mode ii: every rank feeds its GPU from host memory through "
                         "nus_upscaler_upscale_batch at the same time; 0 = skip)")
    ap.add_argument("--schedule", choices=["unit", "three-stage"], default="unit",
                    help="unit: the whole step in one launch of the x2 kernel (default); three-stage: blend, upscale, upscale")
    ap.add_argument("--fused", action="store_true",
                    help="blend inside the second upscale's row loads (in-between frame never written to HBM); "
                         "same output frames, reported separately from the default 3-stage step")
    ap.add_argument("--overlap", action="store_true",
                    help="blend on a second stream, concurrent with the upscale of the real frames (measured: no gain, "
                         "the Lanczos kernel is SIMD-time bound and slows by what the blend takes)")
    ap.add_argument("--config3-seconds", type=float, default=3.0,
                    help="length of the BASELINE-config-3 leg (the plain 300-frame Lanczos stream, back to back, per pattern; "
                         "0 = skip; N=1 only)")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the rank to its GPU's NUMA node")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: create the communicator all the same and issue every collective of the N > 1 path on it "
                         "(the LUT broadcast, the barriers, the gathers): how a one-GPU box rehearses the RCCL calls of the 8-GPU run")
    ap.add_argument("--force-device", type=int, default=-1,
                    help="rehearsal only: put every rank on this GPU (with --backend gloo on a 1-GPU box)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# N > 1 started directly: launch the ranks as a child process tree (never exec, never touch HIP)
# ---------------------------------------------------------------------------------------------

def launch_ranks(nproc: int, script: str, script_args, timeout=None):
    """The product's launcher (nu_scaler_amd.launch.launch_ranks: `python -m torch.distributed.run` as a CHILD process, never an
    exec; this process stays off the GPU -- importing the package loads neither torch nor libnuscaler_hip.so)."""
    from nu_scaler_amd import launch

    return launch.launch_ranks(nproc, script, script_args, timeout=timeout)


def main_launcher(args) -> int:
    rc, lines = launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    for ln in lines:  # rank 0's JSON line on stdout; whatever else the ranks or gloo wrote there goes to stderr
        print(ln, flush=True, file=sys.stdout if ln.startswith("{") else sys.stderr)
    return rc


# ---------------------------------------------------------------------------------------------
# helpers of the worker
# ---------------------------------------------------------------------------------------------

def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file ships with the repo)."""
    fallback = "Mpixels/sec (in+out) at 1080p→4K ×2 upscale + interp, 1/2/4/8 GPU"
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f).get("metric", fallback)
    except (OSError, ValueError):
        return fallback


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def usable_cpus():
    """CPUs this process may actually keep busy (oracle.usable_cpus: the affinity mask capped by the cgroup's CPU quota).
    A GPU box's job gets a share of the host (16 CPUs per GPU on this pool); more OpenMP threads than that burn the quota in
    the first milliseconds of every 100 ms period and sleep through the rest of it (a 1080p Lanczos pass: 4 ms on 64 threads,
    8.4 ms on 16, 100-200 ms on 256 -- profiles/r03_cpu_baseline_threads.txt)."""
    import oracle

    return oracle.usable_cpus()


def cpu_baseline(args, unit_pixels):
    """Time the CPU oracle on the host cores of this box: every algorithm of the path on its own and the
    combined unit, each as the median of 10 runs after 2 warm-ups, single
    thread (how the reference's BasicUpscaler runs) and all cores (OpenMP row-parallel).
    Bounded: a leg whose 12 runs would take more than ~4 s at full size runs on a strip of the frame
    (full width, fewer rows -- every algorithm here is row-separable work); the rate is per pixel."""
    import oracle

    oracle.build()
    w, h = args.width, args.height
    a, b = oracle.gen_gradient(w, h, 0), oracle.gen_gradient(w, h, 1)
    cores = min(oracle.max_threads(), usable_cpus())
    t_start = time.perf_counter()

    def timed(fn_of_rows, runs=10, warm=2, budget_s=4.0):
        """fn_of_rows(rows) runs the leg on the top `rows` rows.  Returns (median seconds, rows used)."""
        t0 = time.perf_counter()
        fn_of_rows(h)
        t_full = time.perf_counter() - t0
        rows = h
        if t_full * (runs + warm) > budget_s:
            rows = max(64, int(h * budget_s / (t_full * (runs + warm))) // 8 * 8)
        for _ in range(warm):
            fn_of_rows(rows)
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn_of_rows(rows)
            ts.append(time.perf_counter() - t0)
        return _median(ts), rows

    legs = {  # name -> (callable(rows, threads), in+out pixels per input row)
        "nearest": (lambda r, th: oracle.nearest(a[:r], 2 * w, 2 * r, threads=th), 5 * w),
        "bilinear": (lambda r, th: oracle.bilinear(a[:r], 2 * w, 2 * r, threads=th), 5 * w),
        "lanczos3": (lambda r, th: oracle.lanczos3(a[:r], 2 * w, 2 * r, threads=th), 5 * w),
        "warp_blend_zero_flow": (lambda r, th: oracle.warp_blend(a[:r], b[:r], None, 0.5, threads=th), 3 * w),
    }
    per_alg = {}
    for name, (fn, pix_per_row) in legs.items():
        t1, r1 = timed(lambda r: fn(r, 1))
        tn, rn = timed(lambda r: fn(r, cores)) if cores > 1 else (t1, r1)
        per_alg[name] = {"Mpix_per_s_1_thread": round(pix_per_row * r1 / t1 / 1e6, 1), "rows_1_thread": r1,
                         "Mpix_per_s_all_cores": round(pix_per_row * rn / tn / 1e6, 1), "rows_all_cores": rn}

    def unit(r, th):
        mid = oracle.warp_blend(a[:r], b[:r], None, 0.5, threads=th)
        oracle.lanczos3(a[:r], 2 * w, 2 * r, threads=th)
        oracle.lanczos3(mid, 2 * w, 2 * r, threads=th)

    unit_pix_per_row = unit_pixels / h
    t1, r1 = timed(lambda r: unit(r, 1), budget_s=6.0)
    tn, rn = timed(lambda r: unit(r, cores), budget_s=6.0) if cores > 1 else (t1, r1)
    return {
        "value": round(unit_pix_per_row * r1 / t1 / 1e6, 3),
        "unit": "Mpix/s",
        "cores": 1,
        "kind": "port",
        "sample": f"the unit of the same stream (zero-flow blend + 2x Lanczos-3 x2) on the top {r1} rows of a {w}x{h} "
                  f"frame pair, median of 10 runs after 2 warm-ups, {t1 * 1e3:.1f} ms per run; oracle/nus_oracle.c, "
                  f"gcc -O2 -ffp-contract=off; per_algorithm: each algorithm alone, same protocol; "
                  f"whole leg {time.perf_counter() - t_start:.1f} s",
        "all_cores": {"value": round(unit_pix_per_row * rn / tn / 1e6, 3), "cores": cores, "rows": rn,
                      "what": "OpenMP row blocks on the CPUs this job may keep busy (affinity mask capped by the cgroup's CPU quota); "
                              f"the host has {os.cpu_count()} hardware threads"},
        "per_algorithm": per_alg,
    }


def measure_traffic(frames_per_launch, unit=True):
    """HBM bytes per launch of the dominant launch (main kernel + its edge-column passes) from the L2's memory-side
    counters, in two separate rocprofv3 --pmc passes over a kernel-only child process (never combined with
    tracing).  gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE
    reports half the bytes of a wide coalesced read (x2); WRITE_SIZE is exact; both in KiB.
    unit: the one-launch step (1 main + 2 edge dispatches per step) or the plain upscale (1 + 1).
    Returns (bytes_per_launch_scaled_to_frames_per_launch, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    n_child, reps = 64, 2
    script = "unit_only.py" if unit else "lanczos_only.py"
    edges_per_main = 2 if unit else 1
    vals = {}
    tmp = tempfile.mkdtemp(prefix="nus_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "tools", script), str(n_child), str(reps)]
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=tmp)
            if res.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} failed (rc {res.returncode})"
            rows = {"main": [], "edges": []}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != counter:
                        continue
                    if "k_lanczos3_x2<" in r["Kernel_Name"]:
                        rows["main"].append(float(r["Counter_Value"]))
                    elif "k_lanczos3_x2_edges<" in r["Kernel_Name"]:
                        rows["edges"].append(float(r["Counter_Value"]))
            if not rows["main"]:
                return None, f"no {counter} rows for k_lanczos3_x2"
            vals[counter] = sum(rows["main"]) / len(rows["main"])
            if rows["edges"]:
                vals[counter] += edges_per_main * sum(rows["edges"]) / len(rows["edges"])
    except Exception as e:  # timeouts, missing files: traffic stays null
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per_unit = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 / n_child
    detail = {"FETCH_SIZE_KiB_per_unit": round(vals["FETCH_SIZE"] / n_child, 1),
              "WRITE_SIZE_KiB_per_unit": round(vals["WRITE_SIZE"] / n_child, 1),
              "fetch_correction": 2.0, "child_units_per_launch": n_child,
              "kernels": ("k_lanczos3_x2<.., UNIT> + 2 x k_lanczos3_x2_edges" if unit else "k_lanczos3_x2 + k_lanczos3_x2_edges")}
    return int(per_unit * frames_per_launch), detail


def check_timed_outputs(frames, mid_t, up_real, up_mid, picks, w, h, first_frame=0, threads=0):
    """Compare frames of the TIMED output buffers with the oracle (outside the timed region).  The source frames are
    REGENERATED on the host (the stream is deterministic: frame `first_frame + k` of the global stream, oracle.gen_gradient) and
    the rank's device-resident input is compared with them first, so a rank holding the wrong shard cannot pass; then the
    in-between frame against oracle.warp_blend(frame k, frame k+1) (bit-exact), the upscaled real frame against
    oracle.lanczos3(frame k), the upscaled in-between frame against oracle.lanczos3 of that blend.  Lanczos tolerance as in
    tests/: every sample within 1 LSB and fewer than 0.1 % of the samples different.  Never raises on a mismatch: returns
    (report, summary) -- the caller gathers the summaries of all ranks before anybody gives up (a rank that left early would
    hang the others in the next collective)."""
    import numpy as np

    import oracle
    from nu_scaler_amd.transfer import to_numpy as fetch  # device -> host through nus_download (the product's own road)

    oracle.build()
    th = threads or usable_cpus()
    report = []
    summary = {"ok": 1.0, "input_ok": 1.0, "mid_exact": 1.0, "max_abs_diff": 0.0, "frac_differing": 0.0, "frames": float(len(picks))}
    for k in picks:
        a, b = oracle.gen_gradient(w, h, first_frame + k), oracle.gen_gradient(w, h, first_frame + k + 1)
        same_in = bool(np.array_equal(fetch(frames[k]), a) and np.array_equal(fetch(frames[k + 1]), b))
        report.append({"frame": int(k), "stream_frame": int(first_frame + k), "buffer": "input", "bit_exact": same_in})
        if not same_in:
            summary["input_ok"] = summary["ok"] = 0.0
        mid = oracle.warp_blend(a, b, None, 0.5, threads=th)
        if mid_t is not None:
            same = bool(np.array_equal(fetch(mid_t[k]), mid))
            report.append({"frame": int(k), "buffer": "mid", "bit_exact": same})
            if not same:
                summary["mid_exact"] = summary["ok"] = 0.0
        for name, got_t, src in (("up_real", up_real, a), ("up_mid", up_mid, mid)):
            want = oracle.lanczos3(src, 2 * w, 2 * h, threads=th).astype(np.int16)
            got = fetch(got_t[k]).astype(np.int16)
            d = np.abs(got - want)
            mx, frac = int(d.max()), float((d != 0).mean())
            report.append({"frame": int(k), "buffer": name, "max_abs_diff": mx, "frac_differing": round(frac, 7)})
            summary["max_abs_diff"] = max(summary["max_abs_diff"], float(mx))
            summary["frac_differing"] = max(summary["frac_differing"], frac)
            if mx > 1 or frac >= 1e-3:
                summary["ok"] = 0.0
    return report, summary


def gather_rows(row, world, dist=None, torch=None, comm_dev=None):
    """nu_scaler_amd.stream.gather_rows (every rank's row of numbers on every rank: one all_gather of a float64 vector)."""
    from nu_scaler_amd import stream as S

    return S.gather_rows(row, world, dist, torch, comm_dev)


def spread(rows, key, digits=4):
    from nu_scaler_amd import stream as S

    return S.spread(rows, key, digits)


def pcie_ceiling(torch, dev, n_in, n_out):
    """What hipMemcpyAsync between PINNED host memory and HBM gets on this box for the host path's two frame sizes,
    each direction alone and both at once on two streams (GB/s; the denominator of config.host_path)."""
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h_in = torch.empty(n_in, dtype=torch.uint8, pin_memory=True)
    h_out = torch.empty(n_out, dtype=torch.uint8, pin_memory=True)
    d_in = torch.empty(n_in, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n_out, dtype=torch.uint8, device=dev)
    reps = 30

    def run(h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    d_in.copy_(h_in, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_out.copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    run(True, True)
    t_h2d, t_d2h, t_both = run(True, False), run(False, True), run(True, True)
    return {"h2d_1080p_frame_GBps": round(n_in / t_h2d / 1e9, 2), "d2h_4k_frame_GBps": round(n_out / t_d2h / 1e9, 2),
            "d2h_4k_frame_ms": round(t_d2h * 1e3, 4), "both_directions_ms_per_frame_pair": round(t_both * 1e3, 4),
            "how": "torch pinned tensors, copy_(non_blocking) on two streams, 30 repetitions after a warm-up"}


def host_path_leg(nsc, syn, torch, w, h, device):
    """PCIe-inclusive rate through the trait-shaped host entry points (mode (ii) of BASELINE.md section 3):
    `upscale(&[u8]) -> Vec<u8>` = nus_upscaler_upscale, `upscale_batch` = nus_upscaler_upscale_batch, the persistent ring
    nus_upscaler_stream_*, `interpolate_py` = nus_interp_interpolate, host buffers in and out, next to the box's pinned-copy
    ceiling.  Never `value`."""
    nb = 12
    frames = [syn.gradient_frame(w, h, k).tobytes() for k in range(nb + 1)]
    u = nsc.PyWgpuUpscaler("quality", "lanczos3", device=device)
    u.initialize(w, h, 2 * w, 2 * h)
    ceiling = pcie_ceiling(torch, torch.device("cuda", device), u.input_size, u.output_size)
    out = bytearray(u.output_size)
    for i in range(3):
        u.upscale_into(frames[i], out)
    ts = []
    for i in range(12):
        t0 = time.perf_counter()
        u.upscale_into(frames[i % len(frames)], out)
        ts.append(time.perf_counter() - t0)
    up_ms = _median(ts) * 1e3
    # upscale_batch: 12 frames per call into caller-owned buffers (pageable, touched before) -- and into pinned ones
    outs = [bytearray(u.output_size) for _ in range(nb)]
    u.upscale_batch_into(frames[:nb], outs)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        u.upscale_batch_into(frames[:nb], outs)
        ts.append((time.perf_counter() - t0) / nb)
    batch_ms = _median(ts) * 1e3
    # the persistent ring: the same frames submitted one at a time, three in flight (nus_upscaler_stream_*)
    u.stream_open()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        tickets = [u.stream_submit(frames[k], outs[k]) for k in range(nb)]
        u.stream_wait(tickets[-1])
        ts.append((time.perf_counter() - t0) / nb)
    u.stream_close()
    stream_ms = _median(ts[1:]) * 1e3
    del outs
    pin_in = torch.empty((nb, u.input_size), dtype=torch.uint8, pin_memory=True)
    pin_out = torch.empty((nb, u.output_size), dtype=torch.uint8, pin_memory=True)
    for k in range(nb):
        pin_in[k] = torch.frombuffer(bytearray(frames[k]), dtype=torch.uint8)
    ins_np, outs_np = [pin_in[k].numpy() for k in range(nb)], [pin_out[k].numpy() for k in range(nb)]
    u.upscale_batch_into(ins_np, outs_np)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        u.upscale_batch_into(ins_np, outs_np)
        ts.append((time.perf_counter() - t0) / nb)
    batch_pinned_ms = _median(ts) * 1e3
    del pin_in, pin_out, ins_np, outs_np
    # the reference's own shapes: fresh `bytes` per result (lib.rs:105-112, :140-154)
    t0 = time.perf_counter()
    for i in range(6):
        o = u.upscale(frames[i])
    fresh_ms = (time.perf_counter() - t0) / 6 * 1e3
    del o
    ts = []
    for _ in range(4):  # each result list is dropped before the next call, as a consumer would: the allocator re-uses the pages
        t0 = time.perf_counter()
        o = u.upscale_batch(frames[:nb])
        ts.append((time.perf_counter() - t0) / nb)
        del o
    fresh_batch_ms = _median(ts[1:]) * 1e3
    it = nsc.WgpuFrameInterpolator(device=device)
    it.interpolate_py(frames[0], frames[1], w, h)
    ts = []
    for i in range(10):
        t0 = time.perf_counter()
        mid = it.interpolate_py(frames[i], frames[i + 1], w, h, time_t=0.5)
        u.upscale_into(frames[i], out)
        u.upscale_into(mid, out)
        ts.append(time.perf_counter() - t0)
    unit_ms = _median(ts) * 1e3
    d2h_ms = ceiling["d2h_4k_frame_ms"]
    return {
        "what": "host bytes in -> host bytes out through nus_upscaler_upscale / nus_upscaler_upscale_batch / "
                "nus_interp_interpolate (PCIe and staging copies included); medians",
        "pcie_ceiling": ceiling,
        "upscale_1080p_to_4k_ms_per_frame": round(up_ms, 3),
        "upscale_frames_per_s": round(1e3 / up_ms, 1),
        "upscale_batch_12_ms_per_frame": round(batch_ms, 3),
        "upscale_batch_frames_per_s": round(1e3 / batch_ms, 1),
        "upscale_batch_vs_single_call": round(up_ms / batch_ms, 3),
        "upscale_batch_frac_of_d2h_ceiling": round(d2h_ms / batch_ms, 3),
        "stream_ring_ms_per_frame": round(stream_ms, 3),
        "stream_ring_frames_per_s": round(1e3 / stream_ms, 1),
        "upscale_batch_12_pinned_buffers_ms_per_frame": round(batch_pinned_ms, 3),
        "upscale_batch_pinned_frac_of_d2h_ceiling": round(d2h_ms / batch_pinned_ms, 3),
        "fresh_bytes_per_result": {"upscale_ms": round(fresh_ms, 3), "upscale_batch_12_ms_per_frame": round(fresh_batch_ms, 3),
                                   "what": "the pyo3 shapes (a new 33 MB bytes object per result, each result dropped before the next call)"},
        "unit_ms": round(unit_ms, 3),
        "unit_source_frames_per_s": round(1e3 / unit_ms, 1),
        "unit_4k_output_frames_per_s": round(2e3 / unit_ms, 1),
        "target_4k_output_frames_per_s": 60,
    }


class ClockSampler:
    """Clocks (sclk, mclk, fclk) and power of THIS rank's GPU while a leg runs, sampled by a thread of this process.
    The GPU is found by its PCI address (placement.query_gpu_pci: HIP's device order is not rocm-smi's on a host where the job
    sees one GPU of eight): first the amdgpu sysfs files of that address (no child process: pp_dpm_sclk / _mclk / _fclk, the
    hwmon power sensor), else a fresh `rocm-smi` child per sample (read-only query; the child never touches HIP) on the index
    `rocm-smi --showbus` lists that address under, else on `-d <fallback_index>`."""

    def __init__(self, bdf=None, period_s=0.6, fallback_index=0):
        import threading

        self.period = period_s
        self.bdf = bdf
        self.fallback_index = fallback_index
        self.samples = []  # (seconds since start, sclk MHz, W, mclk MHz, fclk MHz), None where unreadable
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self.t0 = time.perf_counter()
        self.tool = None
        self.power_cap_W = None

    @staticmethod
    def _parse(text):
        import re

        def clk(name):
            m = re.search(name + r"[^\n]*?\((\d+)\s*Mhz\)", text, re.I)
            return int(m.group(1)) if m else None

        watts = re.search(r"Power \(W\):\s*([0-9.]+)", text)
        return clk("sclk"), (float(watts.group(1)) if watts else None), clk("mclk"), clk("fclk")

    @staticmethod
    def _dpm_current(path):
        """'1: 2100Mhz *' -> 2100 (the level the star marks)."""
        import re

        try:
            with open(path) as f:
                for line in f:
                    if "*" in line:
                        m = re.search(r"(\d+)\s*Mhz", line, re.I)
                        return int(m.group(1)) if m else None
        except OSError:
            pass
        return None

    def _sysfs_paths(self):
        import glob

        if not self.bdf:
            return None
        base = os.path.join("/sys/bus/pci/devices", self.bdf)
        if self._dpm_current(os.path.join(base, "pp_dpm_sclk")) is None:
            return None
        power = None
        for name in ("power1_average", "power1_input"):
            found = glob.glob(os.path.join(base, "hwmon", "hwmon*", name))
            if found:
                power = found[0]
                break
        cap = glob.glob(os.path.join(base, "hwmon", "hwmon*", "power1_cap"))
        try:
            with open(cap[0]) as f:
                self.power_cap_W = round(int(f.read().strip()) / 1e6, 1)
        except (OSError, ValueError, IndexError):
            self.power_cap_W = None
        return {"base": base, "power": power}

    def _sysfs_sample(self, paths):
        watts = None
        if paths["power"]:
            try:
                with open(paths["power"]) as f:
                    watts = round(int(f.read().strip()) / 1e6, 1)
            except (OSError, ValueError):
                pass
        b = paths["base"]
        return (self._dpm_current(os.path.join(b, "pp_dpm_sclk")), watts, self._dpm_current(os.path.join(b, "pp_dpm_mclk")),
                self._dpm_current(os.path.join(b, "pp_dpm_fclk")))

    def _smi_index(self, exe):
        import re

        if not self.bdf:
            return self.fallback_index
        try:
            res = subprocess.run([exe, "--showbus"], capture_output=True, text=True, timeout=15)
            for m in re.finditer(r"GPU\[(\d+)\][^\n]*?PCI Bus:\s*([0-9a-fA-F:.]+)", res.stdout):
                if m.group(2).lower() == self.bdf.lower():
                    return int(m.group(1))
        except Exception:
            pass
        return self.fallback_index

    def _run(self):
        import shutil

        paths = self._sysfs_paths()
        exe = shutil.which("rocm-smi")
        if paths:
            self.tool = f"sysfs {paths['base']}/pp_dpm_{{sclk,mclk,fclk}} + {paths['power'] or 'no power sensor'}"
        elif exe:
            idx = self._smi_index(exe)
            self.tool = f"rocm-smi --showpower --showclocks -d {idx} (PCI {self.bdf or 'unknown'})"
        while (paths or exe) and not self._stop.is_set():
            t = time.perf_counter() - self.t0
            try:
                if paths:
                    smp = self._sysfs_sample(paths)
                else:
                    res = subprocess.run([exe, "--showpower", "--showclocks", "-d", str(idx)], capture_output=True, text=True,
                                         timeout=15)
                    smp = self._parse(res.stdout)
                if any(x is not None for x in smp):
                    self.samples.append((round(t, 2),) + tuple(smp))
            except Exception:
                pass
            self._stop.wait(self.period)

    def start(self):
        self.t0 = time.perf_counter()
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=20)
        return self.samples

    @staticmethod
    def mean(samples, col, digits=0):
        xs = [s[col] for s in samples if s[col] is not None]
        return round(sum(xs) / len(xs), digits) if xs else None


def sustained_leg(torch, do_step, upscaler, seconds, tail_s=3.0, batch=20, bdf=None, device_index=0):
    """The same step for `seconds` of wall time, in batches of `batch` steps with one synchronize each; reports the
    rate over the last `tail_s` seconds (whole batches) next to the rate of the first second, the hipEvent brackets
    of the upscale launches in that tail, and the board's clocks (sclk, mclk, fclk) / power sampled meanwhile."""
    upscaler.set_profiling(True)
    upscaler.profile_collect()
    sampler = ClockSampler(bdf, fallback_index=device_index).start()
    marks = []  # (seconds at the end of the batch, steps so far, launches in the batch, their ms)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(batch):
            do_step()
        torch.cuda.synchronize()
        now = time.perf_counter() - t0
        n += batch
        nl, ms = upscaler.profile_collect()
        marks.append((now, n, nl, ms))
        if now >= seconds:
            break
    samples = sampler.stop()
    upscaler.set_profiling(False)
    total = marks[-1][0]
    first = next(m for m in marks if m[0] >= min(1.0, total))
    k0 = max(i for i, m in enumerate(marks) if m[0] <= max(total - tail_s, 0.0) or i == 0)
    tail = marks[k0 + 1:] or marks[-1:]
    tail_steps = tail[-1][1] - marks[k0][1]
    tail_seconds = tail[-1][0] - marks[k0][0]
    tail_launches = sum(m[2] for m in tail)
    tail_ms = sum(m[3] for m in tail)
    in_tail = [sm for sm in samples if sm[0] >= marks[k0][0]]
    return {
        "seconds": round(total, 2), "steps": n,
        "ms_per_step_first_second": round(first[0] / first[1] * 1e3, 4),
        "ms_per_step_tail": round(tail_seconds / max(tail_steps, 1) * 1e3, 4),
        "tail_seconds": round(tail_seconds, 2), "tail_steps": tail_steps,
        "tail_upscale_launches": tail_launches, "tail_upscale_avg_launch_ms": round(tail_ms / max(tail_launches, 1), 4),
        "clock_power_samples": {"tool": sampler.tool, "power_cap_W": sampler.power_cap_W, "t_s__sclk_MHz__W__mclk_MHz__fclk_MHz": samples,
                                "tail_mean_sclk_MHz": ClockSampler.mean(in_tail, 1), "tail_mean_W": ClockSampler.mean(in_tail, 2, 1),
                                "tail_mean_mclk_MHz": ClockSampler.mean(in_tail, 3), "tail_mean_fclk_MHz": ClockSampler.mean(in_tail, 4)},
    }


def box_calibration(nsc, torch, dev, stream, scratch_src, scratch_dst):
    """The denominators of this box (an on-box copy ceiling next to the 8 TB/s spec figure), each ONE plain
    launch of nus_probe_device on the launch stream between two events, median of 5 after a warm-up: hipMemcpyDtoDAsync and a
    16-B-per-lane stream copy over >= 2 GB (read + written bytes counted), a write-only and a read-only stream, the 1 R : 4 W
    mix of a x2 upscale, and the f32 FMA rate of VGPR-only v_fmac chains at 8 waves per SIMD (nothing touches memory).
    scratch_src / scratch_dst: device tensors the probes may overwrite (the bench's own output buffers, after the check)."""
    L = nsc._capi.lib()
    src, dst = scratch_src.reshape(-1), scratch_dst.reshape(-1)
    copy_bytes = min(src.numel(), dst.numel(), 4 << 30) // 4096 * 4096
    mix_bytes = min(src.numel(), dst.numel() // 4, 2 << 30) // 4096 * 4096
    valu_iters = 20000

    def timed(kind, nbytes, iters=0, reps=5):
        ts = []
        for i in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if L.nus_probe_device(kind, src.data_ptr(), dst.data_ptr(), nbytes, iters, stream or None) != 0:
                raise RuntimeError(nsc._capi.last_error())
            e1.record()
            e1.synchronize()
            if i:
                ts.append(e0.elapsed_time(e1) * 1e-3)
        return _median(ts)

    t_dtod, t_copy = timed(0, copy_bytes), timed(1, copy_bytes)
    t_wr, t_rd = timed(2, copy_bytes), timed(3, copy_bytes)
    t_mix = timed(4, mix_bytes)
    t_valu = timed(5, 0, valu_iters)
    fmas = nsc_probe_valu_lanes() * 16 * valu_iters
    return {
        "hipMemcpyDtoDAsync_GBps": round(2 * copy_bytes / t_dtod / 1e9, 1),
        "stream_copy_float4_GBps": round(2 * copy_bytes / t_copy / 1e9, 1),
        "write_only_GBps": round(copy_bytes / t_wr / 1e9, 1),
        "read_only_GBps": round(copy_bytes / t_rd / 1e9, 1),
        "one_read_four_writes_GBps": round(5 * mix_bytes / t_mix / 1e9, 1),
        "valu_fma_f32_TFLOPs": round(2 * fmas / t_valu / 1e12, 2),
        "bytes": {"copy": copy_bytes, "one_read_four_writes_read": mix_bytes},
        "how": "nus_probe_device (nus_k_probe.hip): one plain launch per figure on the launch stream between two events, median "
               "of 5 after a warm-up; copies count read + written bytes; valu = 2048 blocks x 256 lanes x 16 chains x "
               f"{valu_iters} v_fmac_f32, operands in VGPRs",
    }


def nsc_probe_valu_lanes():
    return 2048 * 256  # kProbeValuBlocks x 256 (nus_kernels.hpp)


def config3_leg(torch, upscaler, frames, up_real, count, stream, seconds, bdf, device_index, up_bytes):
    """BASELINE config 3 by itself: the 300-frame stream through the plain Lanczos-3 x2 kernel, launch after launch, for
    `seconds`; the hipEvent brackets of the last two thirds, clocks and power meanwhile."""
    upscaler.set_profiling(True)
    for _ in range(3):
        upscaler.upscale_device(frames.data_ptr(), up_real.data_ptr(), count, stream)
    torch.cuda.synchronize()
    upscaler.profile_collect()
    sampler = ClockSampler(bdf, fallback_index=device_index).start()
    t0 = time.perf_counter()
    marks = []
    while True:
        for _ in range(20):
            upscaler.upscale_device(frames.data_ptr(), up_real.data_ptr(), count, stream)
        torch.cuda.synchronize()
        nl, ms = upscaler.profile_collect()
        marks.append((time.perf_counter() - t0, nl, ms))
        if marks[-1][0] >= seconds:
            break
    samples = sampler.stop()
    upscaler.set_profiling(False)
    total = marks[-1][0]
    tail = [m for m in marks if m[0] > total / 3.0] or marks[-1:]
    t_start = max([m[0] for m in marks if m[0] <= total / 3.0] or [0.0])
    nl, ms = sum(m[1] for m in tail), sum(m[2] for m in tail)
    in_tail = [s for s in samples if s[0] >= t_start]
    avg = ms / max(nl, 1)
    achieved = up_bytes * count / (avg / 1e3) / 1e9
    return {"seconds": round(total, 2), "launches_in_tail": nl, "avg_launch_ms": round(avg, 4),
            "us_per_frame": round(avg * 1e3 / count, 3), "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "wall_ms_per_launch_tail": round((total - t_start) / max(nl, 1) * 1e3, 4),
            "tail_mean_sclk_MHz": ClockSampler.mean(in_tail, 1), "tail_mean_W": ClockSampler.mean(in_tail, 2, 1),
            "tail_mean_mclk_MHz": ClockSampler.mean(in_tail, 3), "tail_mean_fclk_MHz": ClockSampler.mean(in_tail, 4)}


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------

class _OneLineStdout:
    """A rank's stdout is for ONE JSON line (the driver parses it).  Libraries write there too -- RCCL's version banner when rank 0
    creates its communicator, gloo's "[Gloo] Rank r is connected to n peer ranks" from every rank of an 8-rank job (seen in the
    world-8 rehearsal, tests/test_bench_launcher.py) -- so for the life of the worker file descriptor 1 points at stderr, and
    the line goes to a duplicate of the real stdout taken before anything else ran."""

    def __init__(self):
        sys.stdout.flush()
        self._fd = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line: str) -> None:
        data = (line + "\n").encode()
        while data:
            data = data[os.write(self._fd, data):]


def worker(args):
    stdout_line = _OneLineStdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    device_index = args.force_device if args.force_device >= 0 else local_rank
    nccl = args.backend == "nccl"
    _install_crash_reports()

    # The rank's whole set-up is the product's (nu_scaler_amd.stream.ShardedStream; `python -m nu_scaler_amd.cli stream` runs the
    # same object): onto the CPUs of its GPU's NUMA node BEFORE the first HIP call, host threads sized to the rank's CPU share,
    # the process group, the pipeline, rank 0's LUTs to everyone (RCCL over xGMI), this rank's contiguous shard of the global
    # stream plus the overlap frame, resident in HBM.  bench.py adds the measurement around it.
    import nu_scaler_amd as nsc  # (importing the package loads neither torch nor the library)
    from nu_scaler_amd import stream as S
    from nu_scaler_amd import synthetic as syn

    w, h = args.width, args.height
    n_units = args.units
    schedule = "fused" if args.fused else ("unit" if args.schedule == "unit" and not args.overlap else "three-stage")
    try:
        sh = S.ShardedStream(n_units * world, w, h, source=S.SyntheticSource(args.pattern), backend=args.backend,
                             bind=not args.no_bind, force_device=args.force_device, schedule=schedule,
                             lanczos_mode=args.lanczos_mode, force_collectives=args.force_collectives and world == 1)
    except RuntimeError as e:
        raise SystemExit(f"bench.py: {e}")
    _install_fatal_trace()
    import torch
    import torch.distributed as dist

    place, pipe, dev, comm_dev = sh.placement, sh.pipeline, sh.device, sh.comm_device
    if args.no_bind:
        place.update(bound=False, why_not="--no-bind")
    bdf = place.get("gpu_bdf_by_hip") if place.get("gpu_bdf_verified") is not None else place.get("gpu_bdf")
    my_cpus = max(1, int(place.get("cpus_per_rank") or 1))
    lut_bytes = sh.lut_bytes
    start, count = sh.start, sh.count
    assert count == n_units
    frames, mid, up_real, up_mid = sh.frames, sh.mid, sh.up_real, sh.up_mid
    stream = sh.stream_handle()
    barrier, gather = sh.barrier, sh.gather

    def fill(pattern):
        sh.refill(S.SyntheticSource(pattern))

    unit_schedule = schedule == "unit"

    def do_step():
        if args.overlap:
            pipe.step_overlapped(frames, mid, up_real, up_mid)
        else:
            sh.step()

    profile = not args.no_profile

    def before_timed():  # after the warm-up passes, before the first barrier: the hipEvent brackets of the timed launches
        pipe.upscaler.set_profiling(profile)
        pipe.upscaler.profile_collect()

    elapsed_local = sh.run(args.steps, args.warmup, step=do_step, before_timed=before_timed)

    launches, kernel_ms = pipe.upscaler.profile_collect() if profile else (0, 0.0)
    pipe.upscaler.set_profiling(False)

    # The buffers the timed steps wrote, against the oracle -- on EVERY rank, each on frames of ITS shard regenerated on the
    # host, before any other leg overwrites them.  Nobody raises before the summaries have been gathered.
    checkable = not args.no_check and args.pattern == "gradient" and args.lanczos_mode == "fma"
    check_report, check = None, {"ok": None, "input_ok": None, "mid_exact": None, "max_abs_diff": None, "frac_differing": None,
                                 "frames": None}
    if checkable:
        picks = sorted({0, count // 2 - 1 if count > 2 else 0, count - 1}) if world == 1 else sorted({0, count - 1})
        check_report, check = check_timed_outputs(frames, None if args.fused else mid, up_real, up_mid, picks, w, h,
                                                  first_frame=start, threads=my_cpus)
    else:
        # cheap sanity check: the outputs are fully written (alpha of the opaque stream stays 255)
        assert args.pattern != "gradient" or (int(up_real[0, ..., 3].min()) == 255 and int(up_mid[count - 1, ..., 3].min()) == 255)

    # What the very same step sustains for seconds, on every rank at the same time (never `value`): ms per step over the last
    # 3 s, hipEvent brackets of the upscale launches in that tail, the rank's own GPU's clocks and power sampled meanwhile.
    sustained = None
    if args.sustained_seconds > 0:
        barrier()
        sustained = sustained_leg(torch, do_step, pipe.upscaler, args.sustained_seconds, bdf=bdf, device_index=device_index)
        barrier()

    unit_launch_bytes = pipe.unit_bytes if unit_schedule else (w * h + 4 * w * h) * 4
    cps = (sustained or {}).get("clock_power_samples", {})
    row = {
        "elapsed_s": elapsed_local,
        "bracket_ms": (kernel_ms / launches) if launches else None,
        "frac": (unit_launch_bytes * count / (kernel_ms / launches / 1e3) / 1e9 / HBM_PEAK_GBPS) if launches else None,
        "check_ok": check["ok"], "check_input_ok": check["input_ok"], "check_mid_exact": check["mid_exact"],
        "check_max_abs_diff": check["max_abs_diff"], "check_frac_differing": check["frac_differing"], "check_frames": check["frames"],
        "sustained_ms_per_step": (sustained or {}).get("ms_per_step_tail"),
        "sustained_bracket_ms": (sustained or {}).get("tail_upscale_avg_launch_ms") or None,
        "sclk_MHz": cps.get("tail_mean_sclk_MHz"), "mclk_MHz": cps.get("tail_mean_mclk_MHz"),
        "fclk_MHz": cps.get("tail_mean_fclk_MHz"), "W": cps.get("tail_mean_W"),
        "numa_node": place.get("numa_node"), "bound": 1.0 if place.get("bound") else 0.0, "cpus_per_rank": my_cpus,
        "bdf_verified": None if place.get("gpu_bdf_verified") is None else (1.0 if place["gpu_bdf_verified"] else 0.0),
        "n_cpus_in_mask": place.get("n_cpus_in_mask"), "copy_threads": place.get("copy_threads"),
        "first_frame": start,
    }
    rows = gather(row)
    per_rank = [r["elapsed_s"] for r in rows]
    elapsed = max(per_rank)
    check_failed = checkable and any(r["check_ok"] != 1.0 for r in rows)

    solo = rank == 0 and world == 1
    extras = solo and not args.no_extras and not args.fused and not args.overlap

    def timed_leg(step_fn, n, with_profile=True):
        """(ms per step, bracketed launches, their summed ms) of n steps after 2 warm-ups; informational legs only."""
        for _ in range(2):
            step_fn()
        torch.cuda.synchronize()
        pipe.upscaler.set_profiling(profile and with_profile)
        pipe.upscaler.profile_collect()
        tl = time.perf_counter()
        for _ in range(n):
            step_fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tl) / n * 1e3
        nl, kms = pipe.upscaler.profile_collect() if profile and with_profile else (0, 0.0)
        pipe.upscaler.set_profiling(False)
        return ms, nl, kms

    n_leg = max(3, min(args.steps, 450) // 6, 25 if solo else 3)
    step3 = lambda: pipe.step(frames, mid, up_real, up_mid, stream)       # noqa: E731
    stepu = lambda: pipe.step_unit(frames, mid, up_real, up_mid, stream)  # noqa: E731
    other_schedule_leg = fused_leg = None
    if extras:
        # Informational: the other schedule on the same frames.  With the unit schedule timed, this is rounds 1-2's
        # blend / upscale / upscale sequence, whose bracket holds k_lanczos3_x2 ALONE (+ its edge pass): 41 472 000 B per frame.
        other_schedule_leg = timed_leg(step3 if unit_schedule else stepu, n_leg)
        # Informational: the two 4K outputs only (blend inside the second upscale's row loads, in-between frame not written)
        fused_leg = timed_leg(lambda: pipe.step_fused(frames, up_real, up_mid, stream), n_leg, with_profile=False)

    # The denominators of THIS box, on every rank (a few milliseconds of plain kernels over the output buffers, which the
    # check is done with): copy ceilings and the f32 FMA rate -- boxes of one pool differ by 7-15 % on one binary.
    calib = None
    if not args.no_extras and profile:
        calib = box_calibration(nsc, torch, dev, stream, up_mid, up_real)
    crow = {k: (calib or {}).get(k) for k in ("hipMemcpyDtoDAsync_GBps", "stream_copy_float4_GBps", "write_only_GBps",
                                              "read_only_GBps", "one_read_four_writes_GBps", "valu_fma_f32_TFLOPs")}
    crows = gather(crow)

    # Informational: the on-box ceiling for an upscale's very bytes AND access pattern -- k_nearest_x2 reads the same 1080p
    # frames and writes the same 4K frames with no arithmetic.
    copy_ms = None
    if extras and profile:
        nn = nsc.PyWgpuUpscaler("quality", "nearest", device=device_index)
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

    # Informational: the motion-compensated step (per-pair pyramid + Horn-Schunck flow feeding the warp)
    motion_ms = None
    if extras:
        flows = torch.empty((count, h, w, 2), dtype=torch.float32, device=dev)
        pipe.interp.set_mode("fma")  # the dense-flow warp in its product mode (+-1 LSB, as Lanczos); zero flow is exact either way
        motion = {}
        # the flow front end in its product mode (flow within 1e-3 px), stage after stage and as a two-stream pipeline over chunks
        # of 100 units (the estimator of chunk i+1 beside warp + upscales of chunk i), and in its verification mode
        for name, fmode, piped, fmt in (("fast_pipelined", "fast", True, None), ("fast_pipelined_f32_handoff", "fast", True, "f32"),
                                         ("fast", "fast", False, None), ("exact", "exact", False, None)):
            # the pipeline through ONE entry point (nus_flow_interpolate_device_stream: frames in, in-between frames out; the flows stay
            # in the estimator's workspace, as Rg16Float unless "f32" is asked for); the stage-after-stage legs store f32 flows as
            # rounds 2-4 did
            kw = dict(flow_mode=fmode, pipelined=piped, fused_warp=piped, flow_format=fmt)
            fl = None if piped else flows
            pipe.step_motion(frames, fl, mid, up_real, up_mid, stream, **kw)
            torch.cuda.synchronize()
            tm = time.perf_counter()
            for _ in range(2):
                pipe.step_motion(frames, fl, mid, up_real, up_mid, stream, **kw)
            torch.cuda.synchronize()
            motion[name] = (time.perf_counter() - tm) / 2 * 1e3
        motion_ms = motion["fast_pipelined"]  # ONE named configuration: the product's default; the others are listed beside it
        pipe.interp.set_mode("exact")
        del flows

    up_bytes = (w * h + 4 * w * h) * 4  # algorithmic bytes of one upscaled frame
    # BASELINE config 3 by itself, this pattern: the plain Lanczos stream back to back for seconds
    config3 = {}
    if extras and profile and args.config3_seconds > 0:
        config3[args.pattern] = config3_leg(torch, pipe.upscaler, frames, up_real, count, stream, args.config3_seconds, bdf,
                                            device_index, up_bytes)

    # Informational: both schedules on the other pattern (gradient = opaque frames, the kernel's
    # 3-channel path; noise = real alpha, its 4-channel path), each with its own hipEvent bracket.
    other = None
    if extras:
        other_pattern = "noise" if args.pattern == "gradient" else "gradient"
        fill(other_pattern)
        other = (other_pattern, timed_leg(do_step, n_leg), timed_leg(step3 if unit_schedule else stepu, n_leg))
        if profile and args.config3_seconds > 0:
            config3[other_pattern] = config3_leg(torch, pipe.upscaler, frames, up_real, count, stream, args.config3_seconds, bdf,
                                                 device_index, up_bytes)

    # (the device buffers stay allocated meanwhile: on this ROCm stack, copies in both directions at once take 1.8x as long
    # in a process that has just returned 25 GB to the driver -- profiles/r03_host_path_process_state.txt)
    host_path = None
    if extras:
        host_path = host_path_leg(nsc, syn, torch, w, h, device_index)

    # Mode (ii): every rank feeds its own GPU from host memory at the same time (any N; never `value`).  After the other host
    # legs: the same place in the process's life as config.host_path's batch figure
    host_fed = None
    if args.host_fed_seconds > 0 and not args.no_extras:
        rate = sh.run_host_fed(args.host_fed_seconds)  # (the product's: nu_scaler_amd.stream.ShardedStream.run_host_fed)
        hrows = gather({"rate": rate, "numa_node": place.get("numa_node"), "bound": 1.0 if place.get("bound") else 0.0})
        rates = [r["rate"] for r in hrows]
        frame_mb = (w * h + 4 * w * h) * 4 / 1e6
        by_node = {}
        for r in hrows:
            key = "unknown" if r["numa_node"] is None else str(int(r["numa_node"]))
            by_node[key] = round(by_node.get(key, 0.0) + r["rate"] * frame_mb / 1e3, 2)
        host_fed = {
            "what": "mode (ii): each rank pushes a host-resident shard (12 pageable 1080p frames per call, outputs into pageable "
                    "4K buffers) through nus_upscaler_upscale_batch -- one submitting thread + a retiring thread + the copy pool, "
                    "three slot streams per GPU -- all ranks at the same time, between two barriers",
            "seconds": args.host_fed_seconds, "frames_4k_per_s_total": round(sum(rates), 1),
            "frames_4k_per_s_per_gpu": {"min": round(min(rates), 1), "max": round(max(rates), 1),
                                        "by_rank": [round(x, 1) for x in rates]},
            "host_device_GBps_total": round(sum(rates) * frame_mb / 1e3, 2),
            "host_device_GBps_by_numa_node": by_node,
            "fed_from_numa_node_by_rank": [None if r["numa_node"] is None else int(r["numa_node"]) for r in hrows],
            "bound_by_rank": [bool(r["bound"]) for r in hrows],
            "target_4k_frames_per_s_per_gpu": 60}

    if rank == 0:
        total_units = n_units * world * args.steps
        value = total_units * pipe.unit_pixels / elapsed / 1e6
        # what one bracketed launch processes: a whole unit per frame (107 827 200 B at 1080p: blend 24 883 200 + 2 x 41 472 000,
        # SURVEY.md section 8d) in the unit schedule, one upscaled frame otherwise
        launch_bytes = lambda is_unit: (pipe.unit_bytes if is_unit else up_bytes) * count  # noqa: E731

        def roof(nl, kms, is_unit):
            achieved = launch_bytes(is_unit) / (kms / 1e3 / nl) / 1e9
            return {"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4), "avg_launch_ms": round(kms / nl, 4)}

        roofline = None
        if launches:
            r = roof(launches, kernel_ms, unit_schedule)
            roofline = {
                "bound": "hbm",
                "kernel": ("k_lanczos3_x2<FMA, blend-on-load, UNIT> (+ 2 x k_lanczos3_x2_edges): the whole step in one launch"
                           if unit_schedule else "k_lanczos3_x2 (+ k_lanczos3_x2_edges)"),
                "achieved": r["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": r["frac"], "traffic": None,
                "frac_basis": "SURVEY 8d algorithmic bytes, un-fused", "moved_frac": None,
                "bytes_per_launch": launch_bytes(unit_schedule), "units_per_launch": count, "launches": launches,
                "avg_launch_ms": r["avg_launch_ms"], "pattern": args.pattern,
                "frac_by_rank": spread(rows, "frac"),
                "note": ("achieved = algorithmic bytes of the units one launch processes (SURVEY.md 8d: 107 827 200 B per unit = "
                         "zero-flow interpolation 24 883 200 + two upscales of 41 472 000) / hipEvent time on the launch stream "
                         "inside the timed region; the bracket holds every launch of the step (the unit kernel and the two "
                         "edge-column passes: rocprofv3 --stats shows three kernels).  The unit kernel moves fewer bytes than "
                         "that -- `traffic`, and `moved` = traffic / the same time: the fraction is earned by NOT moving bytes "
                         "(each input row reaches the kernel from HBM about once, not four times; the in-between frame is "
                         "written, never re-read); `upscale_kernel_alone` is k_lanczos3_x2 by itself. "
                         if unit_schedule else
                         "achieved = algorithmic bytes (8 294 400 R + 33 177 600 W per frame) / hipEvent time on the launch "
                         "stream inside the timed region; the bracket holds the main kernel and its edge-column pass. ") +
                        "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 of the bracket's kernels from separate rocprofv3 --pmc "
                        "passes, scaled to units_per_launch",
            }
            if sustained and sustained["tail_upscale_launches"]:
                sr = roof(sustained["tail_upscale_launches"],
                          sustained["tail_upscale_avg_launch_ms"] * sustained["tail_upscale_launches"], unit_schedule)
                sr["what"] = "the same bracket over the last seconds of config.sustained (after the timed region)"
                roofline["sustained"] = sr
            if other_schedule_leg and other_schedule_leg[1]:
                key = "upscale_kernel_alone" if unit_schedule else "unit_kernel"
                roofline[key] = roof(other_schedule_leg[1], other_schedule_leg[2], not unit_schedule)
                roofline[key]["what"] = ("k_lanczos3_x2 + its edge pass by itself (the upscale launches of the three-stage schedule on "
                                         "the same frames, after the timed region): 41 472 000 algorithmic bytes per frame"
                                         if unit_schedule else "the one-launch step on the same frames, after the timed region")
            if calib or copy_ms:
                cc = dict(calib or {})
                if copy_ms:
                    cc["k_nearest_x2_GBps"] = round(up_bytes * count / (copy_ms / 1e3) / 1e9, 1)
                    cc["GBps"] = cc["k_nearest_x2_GBps"]  # (the key rounds 1-3 reported)
                    cc["k_nearest_x2_how"] = ("k_nearest_x2 over the same frames (an upscale's bytes in and out in its own access "
                                              "pattern, no arithmetic), hipEvent time")
                cc["by_rank"] = {k: spread(crows, k, 2) for k in crow} if world > 1 else None
                roofline["copy_ceiling"] = cc
            if config3:
                roofline["config3"] = dict(
                    config3,
                    what="BASELINE config 3 by itself: the 300-frame stream through k_lanczos3_x2 (+ its edge pass), launch after "
                         "launch for seconds, nothing in between; hipEvent brackets of the last two thirds; 41 472 000 "
                         "algorithmic bytes per frame against the 8 TB/s peak")
            if other and other[1][1]:
                roofline["other_pattern"] = dict(roof(other[1][1], other[1][2], unit_schedule), pattern=other[0])
                if other[2][1]:
                    roofline["other_pattern"]["upscale_kernel_alone" if unit_schedule else "unit_kernel"] = roof(
                        other[2][1], other[2][2], not unit_schedule)

        def leg(ms):
            return {"ms_per_step": round(ms, 4), "Mpix_per_s_per_gpu": round(n_units * pipe.unit_pixels / ms / 1e3, 1)}

        per_rank_check = None
        if checkable:
            per_rank_check = {
                "tolerance": "source frames regenerated on the host and equal to the rank's device-resident input; in-between "
                             "frames bit-exact; 4K frames max |diff| <= 1 LSB and < 0.1 % of samples differing (Lanczos FMA mode)",
                "all_ok": not check_failed,
                "by_rank": [{"rank": i, "ok": r["check_ok"] == 1.0, "first_stream_frame": int(r["first_frame"]),
                             "frames_checked": int(r["check_frames"] or 0), "input_ok": r["check_input_ok"] == 1.0,
                             "mid_bit_exact": r["check_mid_exact"] == 1.0, "max_abs_diff": r["check_max_abs_diff"],
                             "frac_differing": r["check_frac_differing"]} for i, r in enumerate(rows)]}
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
                            f"(1 in-between frame per source frame, written to HBM) + Lanczos-3 x2 of real and in-between "
                            f"frame to {2 * w}x{2 * h}",
                "units_per_step_per_gpu": n_units, "pixels_per_unit": pipe.unit_pixels,
                "algorithmic_bytes_per_unit": pipe.unit_bytes, "pattern": args.pattern,
                "lanczos_mode": args.lanczos_mode, "kernel_variant": pipe.upscaler.kernel_variant,
                "schedule": "fused: upscale(real) + upscale(blend(A,B)) with the blend in the row loads, 2 launches"
                            if args.fused else
                            "blend on a second HIP stream, concurrent with the upscale of the real frames"
                            if args.overlap else
                            "unit: one launch per step (two waves per strip side by side: upscale(A) | blend(A,B) on load -> "
                            "in-between rows stored -> upscale), + the two edge-column passes"
                            if unit_schedule else "3 stages back to back on one stream",
                "sharding": f"frame-parallel, contiguous shards, {world} rank(s), LUT broadcast {lut_bytes} B over "
                            f"{'RCCL' if nccl else args.backend}",
                "lut_broadcast_bytes": lut_bytes,
                "ms_per_step_by_rank": {"min": round(min(per_rank) / args.steps * 1e3, 4),
                                        "max": round(max(per_rank) / args.steps * 1e3, 4),
                                        "by_rank": [round(x / args.steps * 1e3, 4) for x in per_rank]},
                "placement": {
                    "rank0": place,
                    "numa_node_by_rank": [None if r["numa_node"] is None else int(r["numa_node"]) for r in rows],
                    "bound_by_rank": [bool(r["bound"]) for r in rows],
                    "gpu_pci_address_confirmed_by_hip_by_rank": [None if r["bdf_verified"] is None else bool(r["bdf_verified"]) for r in rows],
                    "cpus_per_rank": [int(r["cpus_per_rank"]) for r in rows],
                    "n_cpus_in_mask_by_rank": [None if r["n_cpus_in_mask"] is None else int(r["n_cpus_in_mask"]) for r in rows],
                    "copy_threads_by_rank": [None if r["copy_threads"] is None else int(r["copy_threads"]) for r in rows],
                    "how": "nu_scaler_amd/placement.py: PCI address of the rank's HIP device from the KFD topology in sysfs (a "
                           "HIP-query child process only if that cannot be read), NUMA node and local CPUs from the PCI device's "
                           "sysfs entry, os.sched_setaffinity to a disjoint run of whole cores before the first HIP call; copy "
                           "pool / OpenMP sized to the rank's share of the CPUs the job may keep busy"},
                "per_rank": {
                    "bracket_ms": spread(rows, "bracket_ms"), "sustained_ms_per_step": spread(rows, "sustained_ms_per_step"),
                    "sustained_bracket_ms": spread(rows, "sustained_bracket_ms"),
                    "sclk_MHz": spread(rows, "sclk_MHz", 0), "mclk_MHz": spread(rows, "mclk_MHz", 0),
                    "fclk_MHz": spread(rows, "fclk_MHz", 0), "W": spread(rows, "W", 1),
                    "what": "every rank's own numbers (all_gather): hipEvent bracket of the timed region, the sustained leg's "
                            "tail on all ranks at the same time, that rank's GPU's clocks and power in that tail"},
                "per_rank_check": per_rank_check,
                "timed_output_check": None if check_report is None else {
                    "tolerance": per_rank_check["tolerance"], "rank": 0, "frames": check_report},
                "timed_region_s": round(elapsed, 3),
                "sustained": sustained,
                ("three_stage_variant" if unit_schedule else "unit_variant"): None if other_schedule_leg is None else dict(
                    leg(other_schedule_leg[0]),
                    what=("the same three outputs from three launches (blend, upscale, upscale), the schedule of rounds 1-2"
                          if unit_schedule else "the same three outputs from the one-launch step") +
                         "; informational, after the timed region"),
                "fused_variant": None if fused_leg is None else dict(
                    leg(fused_leg[0]),
                    what="the two 4K outputs only: upscale + upscale-with-the-blend-on-load, in-between frame not written; "
                         "informational, after the timed region"),
                ("noise" if args.pattern == "gradient" else "gradient") + "_variant": None if other is None else dict(
                    leg(other[1][0]),
                    what=f"the timed schedule on the {other[0]} stream (gradient = opaque frames, the x2 kernel's "
                         f"3-channel path; noise = real alpha, its 4-channel path); informational, after the timed region",
                    other_schedule=leg(other[2][0])),
                "lanczos_stream_alone": None if not config3 else {
                    p: {"ms_per_300_frames": c["avg_launch_ms"], "us_per_frame": c["us_per_frame"],
                        "frames_per_s": round(count / (c["avg_launch_ms"] / 1e3), 1),
                        "Mpix_per_s": round(count * 5 * w * h / (c["avg_launch_ms"] / 1e3) / 1e6, 1),
                        "sclk_MHz": c["tail_mean_sclk_MHz"], "W": c["tail_mean_W"]} for p, c in config3.items()},
                "motion_variant": None if motion_ms is None else {
                    "what": "three-stage step with a dense flow per pair (3-level pyramid, 50 + 10 + 10 Horn-Schunck steps, FAST "
                            "arithmetic: flow within 1e-3 px of the exact one) feeding the warp (FMA mode) instead of zero flow; "
                            "ms_per_step is ONE configuration, FramePipeline.step_motion's default: the two-stream pipeline over "
                            "150-unit chunks, estimator + warp through one entry point with the flow handed over as Rg16Float, the "
                            "real frames' upscale first on the second stream; beside it the same with an f32 hand-off, stage after "
                            "stage (f32 flows stored), and stage after stage with the bit-exact front end; informational, this rank only",
                    "configuration": "fast flow, pipelined over 150-unit chunks, fused entry point, Rg16Float hand-off",
                    "ms_per_step": round(motion_ms, 3),
                    "units_per_s_per_gpu": round(n_units / motion_ms * 1e3, 1),
                    "pipelined_ms_per_step": round(motion["fast_pipelined"], 3),
                    "pipelined_f32_handoff_ms_per_step": round(motion["fast_pipelined_f32_handoff"], 3),
                    "stage_by_stage_ms_per_step": round(motion["fast"], 3),
                    "exact_flow_ms_per_step": round(motion["exact"], 3)},
                "host_path": host_path,
                "host_fed": host_fed,
                "frames_per_sec_per_gpu_4k_out": round(2 * n_units * args.steps / elapsed, 1),
                "algorithmic_GBps": round(total_units * pipe.unit_bytes / elapsed / 1e9, 1),
            },
            "roofline": roofline,
        }
        if world == 1 and roofline is not None and not args.no_pmc:
            traffic, detail = measure_traffic(count, unit_schedule)
            roofline["traffic"] = traffic
            roofline["traffic_detail"] = detail
            if traffic:
                t_s = roofline["avg_launch_ms"] / 1e3
                moved = traffic / t_s / 1e9
                cc = roofline.get("copy_ceiling") or {}
                # both readings as short scalars next to each other (long strings are cut from the driver's parsed copy):
                # frac = SURVEY 8(d) algorithmic bytes (the un-fused three stages) / time; moved_frac = PMC traffic / the same time
                roofline["moved_frac"] = round(moved / HBM_PEAK_GBPS, 4)
                roofline["moved_GBps"] = round(moved, 1)
                roofline["moved_basis"] = "PMC traffic (2*FETCH_SIZE+WRITE_SIZE) / same bracket"
                roofline["moved"] = {
                    "GBps": round(moved, 1), "frac_of_peak": round(moved / HBM_PEAK_GBPS, 4),
                    "frac_of_copy_ceiling": {k: round(moved / cc[k], 4) for k in
                                             ("stream_copy_float4_GBps", "hipMemcpyDtoDAsync_GBps", "one_read_four_writes_GBps",
                                              "k_nearest_x2_GBps") if cc.get(k)},
                    "what": "the bytes the bracket's kernels really move through HBM (`traffic`, PMC) / the bracket's time: the "
                            "HBM-throughput reading of the same launch, next to `frac` (algorithmic bytes, SURVEY.md 8d)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, pipe.unit_pixels)
        stdout_line.emit(json.dumps(out))
    if world > 1:
        barrier()
    sh.close()  # the process group is the ShardedStream's
    if check_failed:
        bad = [i for i, r in enumerate(rows) if r["check_ok"] != 1.0]
        raise SystemExit(f"bench.py: timed outputs differ from the oracle on rank(s) {bad} (config.per_rank_check)")


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(main_launcher(args))  # parent of the ranks: subprocess only, never the GPU
    worker(args)


if __name__ == "__main__":
    main()
