#!/usr/bin/env python
"""Regenerates the golden fixtures.  Run from the repo root in the BUILD container
(needs /root/reference for the PNG copies; the oracle vectors need only gcc + numpy).

  * ref_*.png are byte copies of image DATA files the reference's own tests hold.
  * oracle_vectors.npz holds seeded inputs and the CPU oracle's outputs for them.
"""
import hashlib
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

COPIES = {
    "ref_test_input.png": "nu_scaler_core/test_input.png",
    "ref_test_output.png": "nu_scaler_core/test_output.png",
    "ref_interp_half.png": "interp_half.png",
}


def main():
    from _png import read_png

    import oracle

    if os.path.isdir(REF):
        for dst, src in COPIES.items():
            shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
            os.chmod(os.path.join(HERE, dst), 0o644)
    with open(os.path.join(HERE, "SHA256SUMS"), "w") as f:
        for dst in sorted(COPIES):
            raw = read_png(os.path.join(HERE, dst)).tobytes()
            f.write(f"{hashlib.sha256(raw).hexdigest()}  {dst} (raw RGBA)\n")

    vec = {}
    noise = oracle.gen_noise(48, 27, 0x5EED)
    vec["noise_48x27"] = noise
    for name, (ow, oh) in {"x2": (96, 54), "x1p5": (72, 41), "down": (20, 11)}.items():
        vec[f"nearest_{name}"] = oracle.nearest(noise, ow, oh)
        vec[f"bilinear_{name}"] = oracle.bilinear(noise, ow, oh)
        vec[f"bilinear_wgsl_{name}"] = oracle.bilinear_wgsl(noise, ow, oh)
        vec[f"lanczos3_{name}"] = oracle.lanczos3(noise, ow, oh)
    a = oracle.gen_noise(40, 24, 1)
    b = oracle.gen_noise(40, 24, 2)
    rng = np.random.default_rng(7)
    flow = (rng.standard_normal((24, 40, 2)) * 3).astype(np.float32)
    vec["warp_a"], vec["warp_b"], vec["warp_flow"] = a, b, flow
    vec["warp_zero_t050"] = oracle.warp_blend(a, b, None, 0.5)
    vec["warp_zero_t030"] = oracle.warp_blend(a, b, None, 0.3)
    vec["warp_flow_t050"] = oracle.warp_blend(a, b, flow, 0.5)
    vec["warp_flow_t025"] = oracle.warp_blend(a, b, flow, 0.25)
    # "next rows" (SURVEY section 8f): the other resize filters, the x4 and down-scaling shapes, FSR1, the flow front end
    small = oracle.gen_noise(24, 14, 0xBEEF)
    vec["noise_24x14"] = small
    vec["catmullrom_x2"] = oracle.resize(small, 48, 28, oracle.FILTER_CATMULLROM)
    vec["triangle_x1p5"] = oracle.resize(small, 36, 21, oracle.FILTER_TRIANGLE)
    vec["lanczos3_x4"] = oracle.lanczos3(small, 96, 56)
    vec["lanczos3_half"] = oracle.lanczos3(noise, 24, 13)
    vec["catmullrom_third"] = oracle.resize(noise, 16, 9, oracle.FILTER_CATMULLROM)
    # round 5: the small rational factors of the P/Q register-window kernel (x6/5, x7/5, x5/4, x5/3, x5/2)
    pq = oracle.gen_noise(40, 15, 0xD1CE)
    vec["noise_40x15"] = pq
    vec["lanczos3_x6o5"] = oracle.lanczos3(pq, 48, 18)
    vec["catmullrom_x7o5"] = oracle.resize(pq, 56, 21, oracle.FILTER_CATMULLROM)
    pq2 = oracle.gen_noise(36, 12, 0xFACE)
    vec["noise_36x12"] = pq2
    vec["lanczos3_x5o3"] = oracle.lanczos3(pq2, 60, 20)
    vec["triangle_x5o3"] = oracle.resize(pq2, 60, 20, oracle.FILTER_TRIANGLE)
    pq3 = oracle.gen_noise(32, 12, 0xCAFE)
    vec["noise_32x12"] = pq3
    vec["lanczos3_x5o4"] = oracle.lanczos3(pq3, 40, 15)
    vec["lanczos3_x5o2"] = oracle.lanczos3(pq3, 80, 30)
    vec["fsr1_x2"] = oracle.fsr1(small, 48, 28, 0.0, 0.7)
    vec["flow_l2_c5_r2"] = oracle.flow_estimate(a, b, 2, 5, 2, 0.02 ** 2)  # lambda: FlowEstimator's default
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **vec)
    print("wrote", len(vec), "arrays")


if __name__ == "__main__":
    main()
