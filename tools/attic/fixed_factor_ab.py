#!/usr/bin/env python
"""A/B of library variants (tools/build_lz_variants.sh) on one fixed-factor Lanczos-3 resize: us per frame for a batch, gradient and
noise input, several rows-per-wave settings.  One child process per (variant, repetition), alternating.
usage: fixed_factor_ab.py iw ih ow oh frames variant[,variant...] [rows_per_wave,...]     (variant "-" = the product library)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def worker(iw, ih, ow, oh, n, ths):
    sys.path.insert(0, ROOT)
    import torch
    import nu_scaler_amd as nsc
    from nu_scaler_amd import synthetic as syn
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    out = torch.empty((n, oh, ow, 4), dtype=torch.uint8, device=dev)
    res = []
    warm = nsc.PyWgpuUpscaler("quality", os.environ.get("NUS_AB_ALG", "lanczos3"))  # clocks and caches settle before the first timed configuration
    warm.initialize(iw, ih, ow, oh)
    frames = syn.gradient_stream_torch(n, iw, ih, dev)
    for _ in range(20):
        warm.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
    torch.cuda.synchronize()
    for pat, gen in (("gradient", syn.gradient_stream_torch), ("noise", syn.noise_stream_torch)):
        frames = gen(n, iw, ih, dev)
        for th in ths:
            u = nsc.PyWgpuUpscaler("quality", os.environ.get("NUS_AB_ALG", "lanczos3"))
            if th:
                u.set_option("rows_per_wave", th)
            u.initialize(iw, ih, ow, oh)
            for _ in range(3):
                u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    u.upscale_device(frames.data_ptr(), out.data_ptr(), n, s)
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) / 5 / n * 1e3)
            res.append(f"{pat[:4]} th={th}: {best:6.2f}")
    print(f"{u.kernel_variant}  " + "   ".join(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        a = sys.argv[2:]
        worker(*(int(v) for v in a[:5]), [int(v) for v in a[5].split(",")])
        sys.exit(0)
    dims = sys.argv[1:6]
    variants = sys.argv[6].split(",")
    ths = sys.argv[7] if len(sys.argv) > 7 else "0"
    for rep in range(2):
        for v in variants:
            env = dict(os.environ)
            if v != "-":
                env["NUS_LIB_PATH"] = os.path.join(HERE, "_ablate", f"lib_{v}.so")
            r = subprocess.run([sys.executable, __file__, "--worker", *dims, ths], env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if "th=" in l]
            print(f"{v:6s} {line[0] if line else 'FAILED ' + r.stderr[-300:]}", flush=True)
