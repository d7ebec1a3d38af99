"""`python -m nu_scaler_amd.cli` -- the image-file commands of the north star's `nu_scaler_cli`
(SURVEY.md section 8f rank 2): `upscale <in.png> <out.png> --algorithm --scale`, following the
legacy crate's `upscale_image_file` (Nu_scale/src/upscale/mod.rs:307-338) and the option names of
its `fullscreen` subcommand (Nu_scale/src/main.rs:36-73: --tech, --quality, --algorithm).
Everything runs on the HIP device; without one the command fails (no CPU path).
"""
from __future__ import annotations

import argparse
import sys


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="nu_scaler_cli", description="NU_Scaler image tools on the HIP path")
    sub = ap.add_subparsers(dest="command", required=True)
    up = sub.add_parser("upscale", help="upscale a PNG")
    up.add_argument("input")
    up.add_argument("output")
    up.add_argument("--tech", default="fallback", help="fsr, fallback or none (Nu_scale/src/main.rs:46-52)")
    up.add_argument("--quality", default="quality", help="ultra, quality, balanced or performance")
    up.add_argument("--algorithm", default=None,
                    help="nearest, bilinear, bicubic, lanczos3, triangle, fsr1, easu (default: by quality)")
    up.add_argument("--scale", type=float, default=2.0, help="scale factor (output = trunc(input * scale))")
    up.add_argument("--device", type=int, default=0)
    it = sub.add_parser("interpolate", help="in-between frame of two PNGs of equal size")
    it.add_argument("frame_a")
    it.add_argument("frame_b")
    it.add_argument("output")
    it.add_argument("--t", type=float, default=0.5, help="time of the new frame between A (0) and B (1)")
    it.add_argument("--flow", action="store_true", help="estimate motion (pyramid + Horn-Schunck) instead of zero flow")
    it.add_argument("--device", type=int, default=0)
    st = sub.add_parser("stream", help="the sharded frame-queue stream: a synthetic 1080p stream through interpolate + x2 upscale "
                                       "on N GPUs of this node, one process per GPU, frames sharded, LUTs broadcast over RCCL")
    st.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node (started as a child process tree)")
    st.add_argument("--units", type=int, default=60, help="source frames per GPU (weak scaling: the stream has gpus x units)")
    st.add_argument("--width", type=int, default=1920)
    st.add_argument("--height", type=int, default=1080)
    st.add_argument("--steps", type=int, default=5)
    st.add_argument("--warmup", type=int, default=1)
    st.add_argument("--pattern", default="gradient", choices=["gradient", "noise"])
    st.add_argument("--schedule", default="unit", choices=["unit", "three-stage", "fused"])
    st.add_argument("--algorithm", default="lanczos3", help="an exact-x2 resize filter: lanczos3, bicubic, triangle")
    st.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI; gloo: rehearsal")
    st.add_argument("--force-device", type=int, default=-1, help="every rank on this device (rehearsal on a one-GPU box)")
    st.add_argument("--no-bind", action="store_true", help="leave the ranks' CPU affinity alone")
    st.add_argument("--digest", action="store_true", help="also print one digest per unit of the stream (sharding-invariant)")
    st.add_argument("--window", type=int, default=0,
                    help="0: every rank holds its whole shard in HBM and repeats it --steps times; W > 0: ONE pass over the stream in "
                         "windows of W units per rank, the next window fetched beside the current one (a stream that need not fit)")
    st.add_argument("--host-fed-seconds", type=float, default=0.0,
                    help="afterwards, every rank feeds its GPU from HOST memory through upscale_batch for this long (mode ii)")
    return ap


def _stream_in_windows(args, S) -> dict:
    """`stream --window W`: one pass over the stream, every rank's shard in windows (ShardedStream.run_windows)."""
    s = S.ShardedStream(args.units * args.gpus, args.width, args.height, source=S.SyntheticSource(args.pattern), backend=args.backend,
                        bind=not args.no_bind, force_device=args.force_device, schedule=args.schedule, algorithm=args.algorithm,
                        resident=False)
    try:
        digests = []

        def consume(st, first, n, mid, up_real, up_mid):
            if args.digest:
                digests.extend(st.unit_digests((mid, up_real, up_mid), n))

        s.run_windows(args.window, consume if args.digest else None)
        per_rank = -(-s.total_units // s.world)
        row = {"elapsed_s": s.elapsed_local, "first_unit": float(s.start), "units": float(s.count),
               "numa_node": s.placement.get("numa_node"), "bound": 1.0 if s.placement.get("bound") else 0.0}
        if args.digest:
            row.update({f"sink_digest_{k:05d}": (float(digests[k]) if k < len(digests) else None) for k in range(per_rank)})
        rows = s.gather(row)
        out = s.summarize(rows)
        out.update(rank=s.rank, rows=rows, window=args.window,
                   placement={k: v for k, v in s.placement.items() if not k.startswith("_")})
        return out
    finally:
        s.close()


def stream_command(args, argv) -> int:
    """`stream`: started directly with --gpus N > 1 this process launches the N ranks as children (never an exec, never a HIP
    call here) and relays rank 0's JSON line; inside a rank (or with one GPU) it runs the rank's whole job (stream.run_sharded)."""
    import json

    from . import launch

    if args.gpus > 1 and not launch.under_launcher():
        rc, lines = launch.launch_ranks(args.gpus, "nu_scaler_amd.cli", argv, module=True)
        for ln in lines:
            print(ln, flush=True, file=sys.stdout if ln.startswith("{") else sys.stderr)
        return rc
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"nu_scaler_cli: error: --gpus {args.gpus} but WORLD_SIZE is {world}", file=sys.stderr)
        return 2
    from . import stream as S

    def sink(s):
        row = {}
        if args.host_fed_seconds > 0:
            row["host_fed_frames_per_s"] = s.run_host_fed(args.host_fed_seconds, algorithm=args.algorithm)
        if args.digest:
            per_rank = -(-s.total_units // s.world)  # every rank sends the same number of columns
            d = s.unit_digests()
            row.update({f"digest_{k:05d}": (float(d[k]) if k < len(d) else None) for k in range(per_rank)})
        return row or None

    if args.window > 0:
        out = _stream_in_windows(args, S)
    else:
        out = S.run_sharded(args.units * args.gpus, args.width, args.height, steps=args.steps, warmup=args.warmup,
                            source=S.SyntheticSource(args.pattern), sink=sink, backend=args.backend, bind=not args.no_bind,
                            force_device=args.force_device, schedule=args.schedule, algorithm=args.algorithm)
    if out["rank"] == 0:
        rows = out.pop("rows")
        if args.digest:
            digests = []
            for r in rows:
                digests += [int(r[k]) for k in sorted(r) if k.startswith("sink_digest_") and r[k] is not None]
            out["unit_digests"] = digests
        if args.host_fed_seconds > 0:
            rates = [r["sink_host_fed_frames_per_s"] for r in rows]
            out["host_fed_4k_frames_per_s"] = {"total": round(sum(rates), 1), "by_rank": [round(x, 1) for x in rates]}
        out["bound_by_rank"] = [bool(r["bound"]) for r in rows]
        out["numa_node_by_rank"] = [None if r["numa_node"] is None else int(r["numa_node"]) for r in rows]
        print(json.dumps(out), flush=True)
    return 0


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    if args.command == "stream":
        try:
            return stream_command(args, list(sys.argv[1:] if argv is None else argv))
        except (OSError, ValueError, RuntimeError) as e:
            print(f"nu_scaler_cli: error: {e}", file=sys.stderr)
            return 1
    from . import imagefile
    try:
        if args.command == "upscale":
            ow, oh = imagefile.upscale_image_file(args.input, args.output, args.tech, args.quality, args.scale,
                                                  args.algorithm, device=args.device)
            print(f"{args.output}: {ow}x{oh}")
        else:
            w, h = imagefile.interpolate_image_files(args.frame_a, args.frame_b, args.output, args.t, args.flow,
                                                     device=args.device)
            print(f"{args.output}: {w}x{h}")
    except (OSError, ValueError, RuntimeError) as e:
        print(f"nu_scaler_cli: error: {e}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
