"""CPU: the oracle against the reference's own fixtures and against its numpy twin."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_fixture_hashes(golden):
    want = {}
    with open(os.path.join(GOLDEN, "SHA256SUMS")) as f:
        for line in f:
            h, name = line.split()[:2]
            want[name] = h
    for key, name in (("test_input", "ref_test_input.png"), ("test_output", "ref_test_output.png"),
                      ("interp_half", "ref_interp_half.png")):
        assert hashlib.sha256(golden[key].tobytes()).hexdigest() == want[name]


def test_input_is_reference_py_gradient(golden):
    from oracle import oracle_np as onp

    assert golden["test_input"].shape == (240, 320, 4)
    assert np.array_equal(onp.gen_py_gradient(320, 240), golden["test_input"])


@pytest.mark.parametrize("form", ["bilinear", "bilinear_wgsl"])
def test_bilinear_pinned_by_reference_output(oracle_mod, golden, form):
    """ref_test_output.png: output pixels x<320, y<240 are the reference's bilinear x2
    (the rest of that image is zero: old dispatch bug)."""
    out = getattr(oracle_mod, form)(golden["test_input"], 640, 480)
    gold = golden["test_output"]
    assert np.array_equal(out[:240, :320], gold[:240, :320])
    assert not gold[240:].any() and not gold[:240, 320:].any()


def test_nearest_is_not_what_the_fixture_holds(oracle_mod, golden):
    out = oracle_mod.nearest(golden["test_input"], 640, 480)
    d = np.abs(out[:240, :320].astype(int) - golden["test_output"][:240, :320].astype(int))
    assert d.max() == 1 and (d > 0).sum() == 2560


def test_warp_blend_zero_flow_pinned_by_interp_half(oracle_mod, golden):
    a = oracle_mod.gen_box(64, 64, (255, 0, 0, 255))
    b = oracle_mod.gen_box(64, 64, (0, 0, 255, 255))
    out = oracle_mod.warp_blend(a, b, None, 0.5)
    assert np.array_equal(out, golden["interp_half"])
    assert tuple(out[32, 32]) == (127, 0, 127, 255)  # truncation, not rounding


def test_nearest_x2_is_replication(oracle_mod):
    img = oracle_mod.gen_noise(37, 21)
    out = oracle_mod.nearest(img, 74, 42)
    assert np.array_equal(out, np.repeat(np.repeat(img, 2, axis=0), 2, axis=1))


@pytest.mark.parametrize("dims", [(96, 54), (72, 41), (20, 11), (48, 27), (49, 28)])
def test_c_oracle_matches_numpy_twin(oracle_mod, dims):
    from oracle import oracle_np as onp

    img = oracle_mod.gen_noise(48, 27)
    ow, oh = dims
    assert np.array_equal(oracle_mod.nearest(img, ow, oh), onp.nearest(img, ow, oh))
    assert np.array_equal(oracle_mod.bilinear(img, ow, oh), onp.bilinear(img, ow, oh))
    assert np.array_equal(oracle_mod.bilinear_wgsl(img, ow, oh), onp.bilinear_wgsl(img, ow, oh))
    # numpy's float32 sin may differ from libm's by an ulp: allow +-1 on a few samples
    d = np.abs(oracle_mod.lanczos3(img, ow, oh).astype(int) - onp.lanczos3(img, ow, oh).astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_c_warp_matches_numpy_twin(oracle_mod):
    from oracle import oracle_np as onp

    a = oracle_mod.gen_noise(40, 24, 1)
    b = oracle_mod.gen_noise(40, 24, 2)
    flow = (np.random.default_rng(3).standard_normal((24, 40, 2)) * 4).astype(np.float32)
    for t in (0.0, 0.25, 0.5, 1.0):
        assert np.array_equal(oracle_mod.warp_blend(a, b, flow, t), onp.warp_blend(a, b, flow, t))
        assert np.array_equal(oracle_mod.warp_blend(a, b, None, t), onp.warp_blend(a, b, None, t))


def test_oracle_vectors_stable(oracle_mod):
    """The committed vectors are what the current oracle produces (guards the oracle)."""
    v = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    noise = v["noise_48x27"]
    assert np.array_equal(noise, oracle_mod.gen_noise(48, 27, 0x5EED))
    for name, (ow, oh) in {"x2": (96, 54), "x1p5": (72, 41), "down": (20, 11)}.items():
        assert np.array_equal(v[f"nearest_{name}"], oracle_mod.nearest(noise, ow, oh))
        assert np.array_equal(v[f"bilinear_{name}"], oracle_mod.bilinear(noise, ow, oh))
        assert np.array_equal(v[f"bilinear_wgsl_{name}"], oracle_mod.bilinear_wgsl(noise, ow, oh))
        assert np.array_equal(v[f"lanczos3_{name}"], oracle_mod.lanczos3(noise, ow, oh))
    a, b, flow = v["warp_a"], v["warp_b"], v["warp_flow"]
    assert np.array_equal(v["warp_zero_t050"], oracle_mod.warp_blend(a, b, None, 0.5))
    assert np.array_equal(v["warp_flow_t025"], oracle_mod.warp_blend(a, b, flow, 0.25))
    # next rows
    small = v["noise_24x14"]
    assert np.array_equal(small, oracle_mod.gen_noise(24, 14, 0xBEEF))
    assert np.array_equal(v["catmullrom_x2"], oracle_mod.resize(small, 48, 28, oracle_mod.FILTER_CATMULLROM))
    assert np.array_equal(v["triangle_x1p5"], oracle_mod.resize(small, 36, 21, oracle_mod.FILTER_TRIANGLE))
    assert np.array_equal(v["lanczos3_x4"], oracle_mod.lanczos3(small, 96, 56))
    assert np.array_equal(v["lanczos3_half"], oracle_mod.lanczos3(noise, 24, 13))
    assert np.array_equal(v["catmullrom_third"], oracle_mod.resize(noise, 16, 9, oracle_mod.FILTER_CATMULLROM))
    # round 5: the P/Q factors
    pq, pq2, pq3 = v["noise_40x15"], v["noise_36x12"], v["noise_32x12"]
    assert np.array_equal(pq, oracle_mod.gen_noise(40, 15, 0xD1CE)) and np.array_equal(pq2, oracle_mod.gen_noise(36, 12, 0xFACE))
    assert np.array_equal(pq3, oracle_mod.gen_noise(32, 12, 0xCAFE))
    assert np.array_equal(v["lanczos3_x6o5"], oracle_mod.lanczos3(pq, 48, 18))
    assert np.array_equal(v["catmullrom_x7o5"], oracle_mod.resize(pq, 56, 21, oracle_mod.FILTER_CATMULLROM))
    assert np.array_equal(v["lanczos3_x5o3"], oracle_mod.lanczos3(pq2, 60, 20))
    assert np.array_equal(v["triangle_x5o3"], oracle_mod.resize(pq2, 60, 20, oracle_mod.FILTER_TRIANGLE))
    assert np.array_equal(v["lanczos3_x5o4"], oracle_mod.lanczos3(pq3, 40, 15))
    assert np.array_equal(v["lanczos3_x5o2"], oracle_mod.lanczos3(pq3, 80, 30))
    assert np.array_equal(v["fsr1_x2"], oracle_mod.fsr1(small, 48, 28, 0.0, 0.7))
    assert np.array_equal(v["flow_l2_c5_r2"], oracle_mod.flow_estimate(a, b, 2, 5, 2, 0.02 ** 2))


def test_lanczos_weights_sum_to_one_and_mirror(oracle_mod):
    left, ntaps, w = oracle_mod.resize_axis(1920, 3840)
    assert np.allclose(w.sum(axis=1), 1.0, atol=1e-6)
    assert ntaps.max() == 7 and left[0] == 0 and left[-1] + ntaps[-1] == 1920
    # interior phases: position independent
    assert np.array_equal(w[8], w[1000]) and np.array_equal(w[9], w[1001])


def test_mt_variants_equal_single_thread(oracle_mod):
    img = oracle_mod.gen_noise(64, 36)
    assert np.array_equal(oracle_mod.lanczos3(img, 128, 72), oracle_mod.lanczos3(img, 128, 72, threads=4))
    assert np.array_equal(oracle_mod.bilinear(img, 128, 72), oracle_mod.bilinear(img, 128, 72, threads=4))
    assert np.array_equal(oracle_mod.nearest(img, 128, 72), oracle_mod.nearest(img, 128, 72, threads=4))
    a, b = img, oracle_mod.gen_noise(64, 36, 9)
    assert np.array_equal(oracle_mod.warp_blend(a, b, None, 0.5), oracle_mod.warp_blend(a, b, None, 0.5, threads=4))


@pytest.mark.parametrize("filt", [0, 1, 2])
def test_many_core_resize_is_bit_identical_to_the_oracle(oracle_mod, filt):
    """bench.py's all-cores CPU baseline (orc_resize_mt / orc_lanczos3_mt: row blocks per thread, per-column weights
    computed once, a per-thread f32 row) against the one-thread oracle loop: same bytes at every size, scale
    direction and thread count."""
    for (iw, ih, ow, oh) in [(64, 36, 128, 72), (97, 41, 194, 82), (120, 50, 77, 31), (33, 20, 100, 57), (48, 48, 48, 48)]:
        img = oracle_mod.gen_noise(iw, ih, 1000 + iw)
        want = oracle_mod.resize(img, ow, oh, filt)
        for threads in (2, 3, 0):
            assert np.array_equal(oracle_mod.resize(img, ow, oh, filt, threads=threads), want), (filt, iw, ih, ow, oh, threads)
    big = oracle_mod.gen_gradient(640, 360, 3)
    assert np.array_equal(oracle_mod.lanczos3(big, 1280, 720, threads=0), oracle_mod.lanczos3(big, 1280, 720))


def test_oracle_mt_entries_under_address_and_ub_sanitizers(tmp_path):
    """Every orc_*_mt entry at 1920x1080 with threads=0 (all cores) and odd thread counts, and ragged small sizes, in a build of
    the oracle with -fsanitize=address,undefined (tests/c_abi/oracle_mt_sanitize.c).  Round 5: orc_warp_blend_mt(.., threads=0)
    at this size was the last native call before round 4's unexplained abort; no report."""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "oracle_mt_asan")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fopenmp", "-std=c11",
           "-I", os.path.join(root, "oracle"), os.path.join(root, "tests", "c_abi", "oracle_mt_sanitize.c"),
           os.path.join(root, "oracle", "nus_oracle.c"), "-lm", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and "san" in (res.stderr or "").lower():
        pytest.skip("libasan / libubsan not installed")
    assert res.returncode == 0, res.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:exitcode=67", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=68")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert run.returncode == 0 and "all ok" in run.stdout and "runtime error" not in run.stderr, run.stdout[-2000:] + run.stderr[-4000:]
