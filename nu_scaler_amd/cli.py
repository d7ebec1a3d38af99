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
    return ap


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
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
