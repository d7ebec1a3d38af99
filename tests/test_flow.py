"""Optical-flow front end ("next" row): oracle sanity on CPU,
HIP kernels against the oracle on GPU.  No reference fixture pins this row."""
import numpy as np
import pytest
from conftest import guarded  # device outputs between poisoned guard bands (tests/conftest.py)
from nu_scaler_amd.transfer import to_device as put, to_numpy as fetch  # host <-> HBM through nus_upload / nus_download, never
# torch's pageable copies (docs/d2h_fault_analysis.md)


def _smooth(w, h, shift=0.0):
    """Smooth textured RGBA8 frame whose content is displaced by `shift` pixels in x."""
    x = np.arange(w, dtype=np.float64)[None, :] - shift
    y = np.arange(h, dtype=np.float64)[:, None]
    v = 127.5 + 45 * np.sin(x / 3.0) * np.cos(y / 4.0) + 50 * np.sin((x + 2 * y) / 23.0) + 25 * np.sin(x / 9.0 + y / 11.0)
    img = np.empty((h, w, 4), np.uint8)
    img[..., 0] = np.clip(v, 0, 255)
    img[..., 1] = np.clip(255 - v, 0, 255)
    img[..., 2] = np.clip(v * 0.5 + 40, 0, 255)
    img[..., 3] = 255
    return img


# ---- CPU: oracle properties -----------------------------------------------------------

def test_oracle_blur_and_downsample_properties(oracle_mod):
    flat = np.full((9, 13, 4), 0.625, np.float32)
    assert np.array_equal(oracle_mod.blur(flat), flat)           # weights sum to exactly 1
    assert np.array_equal(oracle_mod.downsample(flat), np.full((5, 7, 4), 0.625, np.float32))
    img = oracle_mod.rgba8_to_f32(oracle_mod.gen_noise(13, 9, 4))
    assert img.dtype == np.float32 and img.max() <= 1.0 and img.min() >= 0.0
    b = oracle_mod.blur(img)
    # interior pixel equals the separable 5x5 binomial applied in H-then-V order
    k = np.array([1, 4, 6, 4, 1], np.float32) / np.float32(16)
    hpass = sum(img[4, 6 + d - 2] * k[d] for d in range(5))
    assert np.allclose(oracle_mod.blur(img)[4, 6], sum(
        (sum(img[4 + e - 2, 6 + d - 2] * k[d] for d in range(5))) * k[e] for e in range(5)), atol=1e-6)
    d = oracle_mod.downsample(img)
    assert d.shape == (5, 7, 4)
    assert np.allclose(d[1, 2], (img[2, 4] + img[2, 5] + img[3, 4] + img[3, 5]) * 0.25, atol=1e-7)
    assert np.allclose(d[4, 6], (img[8, 12] * 4) * 0.25, atol=1e-7)  # odd edge: clamped reads


def test_oracle_horn_schunck_recovers_a_shift(oracle_mod):
    w, h = 96, 64
    a, b = _smooth(w, h, 0.0), _smooth(w, h, 1.0)  # content moves +1 px in x from A to B
    flow = oracle_mod.flow_estimate(a, b, levels=3, coarse_iters=100, refine_iters=30)
    core = flow[12:-12, 12:-12]
    assert abs(np.median(core[..., 0]) - 1.0) < 0.35 and abs(np.median(core[..., 1])) < 0.2
    # zero motion -> zero flow
    z = oracle_mod.flow_estimate(a, a, levels=3, coarse_iters=20, refine_iters=5)
    assert np.abs(z).max() == 0.0


def test_oracle_upsample_identity_and_scale(oracle_mod):
    f = np.random.default_rng(1).standard_normal((7, 9, 2)).astype(np.float32)
    assert np.allclose(oracle_mod.flow_upsample(f, 9, 7, 1.0), f, atol=1e-6)
    u = oracle_mod.flow_upsample(np.ones((4, 5, 2), np.float32), 10, 8, 2.0)
    assert np.allclose(u, 2.0)


# ---- GPU: kernels vs oracle -------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("size", [(64, 40), (65, 33), (1, 1), (5, 3), (130, 71)])
def test_flow_primitives_match_oracle(nsc, oracle_mod, size):
    w, h = size
    fe = nsc.FlowEstimator()
    u8 = oracle_mod.gen_noise(w, h, 31)
    img = fe.rgba8_to_f32(u8)
    assert np.array_equal(img, oracle_mod.rgba8_to_f32(u8))
    assert np.array_equal(fe.blur(img), oracle_mod.blur(img))
    assert np.array_equal(fe.downsample(img), oracle_mod.downsample(img))
    img2 = oracle_mod.rgba8_to_f32(oracle_mod.gen_noise(w, h, 32))
    rng = np.random.default_rng(w + h)
    f0 = rng.standard_normal((h, w, 2)).astype(np.float32)
    for fin, it in ((None, 1), (None, 4), (f0, 3), (f0, 15), (None, 8)):
        want = oracle_mod.horn_schunck(img, img2, fin, iterations=it, lam=4e-4)
        for tiled in (1, 2, 3, 0):  # multi-step kernel by size, LDS tiles, register-pipelined strips, plain per-step kernel
            fe.set_tiled(tiled)
            got = fe.horn_schunck(img, img2, fin, iterations=it, lambda_=4e-4)
            assert np.array_equal(got, want), (size, it, tiled)
    for (dw, dh, sc) in ((2 * w, 2 * h, 2.0), (2 * w - 1, 2 * h - 1, 2.0), (w, h, 1.0), (3 * w + 1, h + 2, 0.5)):
        assert np.array_equal(fe.upsample(f0, dw, dh, sc), oracle_mod.flow_upsample(f0, dw, dh, sc)), (dw, dh)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(200, 300), (64, 129), (65, 128), (118, 90), (119, 257), (173, 70)])
def test_horn_schunck_streamed_kernel_row_blocks_and_strips(nsc, oracle_mod, size):
    """The register-pipelined kernel forced onto images tall enough for several row blocks (each with a K-row halo
    whose rows are wrong by construction and must never be written) and widths of one strip, two overlapping strips and
    at / one past the width where a third strip appears (118 with five steps per launch)."""
    w, h = size
    fe = nsc.FlowEstimator()
    img = oracle_mod.rgba8_to_f32(oracle_mod.gen_noise(w, h, 5))
    img2 = oracle_mod.rgba8_to_f32(oracle_mod.gen_noise(w, h, 6))
    f0 = np.random.default_rng(w * h).standard_normal((h, w, 2)).astype(np.float32)
    fe.set_tiled(3)
    for fin, it in ((None, 1), (f0, 5), (f0, 7), (None, 12)):
        want = oracle_mod.horn_schunck(img, img2, fin, iterations=it, lam=4e-4)
        assert np.array_equal(fe.horn_schunck(img, img2, fin, iterations=it, lambda_=4e-4), want), (size, it)


@pytest.mark.gpu
def test_flow_estimate_matches_oracle_and_improves_interpolation(nsc, oracle_mod):
    w, h = 192, 108
    a, b = _smooth(w, h, 0.0), _smooth(w, h, 4.0)
    fe = nsc.FlowEstimator(levels=3, coarse_iterations=60, refine_iterations=15)
    flow = fe.estimate(a.tobytes(), b.tobytes(), w, h)
    want = oracle_mod.flow_estimate(a, b, 3, 60, 15, fe.lambda_)
    assert np.array_equal(flow, want)  # fused pyramid kernel + LDS-tiled multi-step Horn-Schunck
    fe.set_tiled(False)               # one plain kernel per shader dispatch
    assert np.array_equal(fe.estimate(a.tobytes(), b.tobytes(), w, h), want)
    fe.set_tiled(True)
    # the estimated flow makes the in-between frame closer to the true half-way frame than zero flow does
    truth = _smooth(w, h, 2.0).astype(np.int16)
    it = nsc.WgpuFrameInterpolator()
    mid_flow = np.frombuffer(it.interpolate_py(a.tobytes(), b.tobytes(), w, h, time_t=0.5, flow=flow), np.uint8).reshape(h, w, 4)
    mid_zero = np.frombuffer(it.interpolate_py(a.tobytes(), b.tobytes(), w, h, time_t=0.5), np.uint8).reshape(h, w, 4)
    core = (slice(16, -16), slice(16, -16))
    err_flow = np.abs(mid_flow.astype(np.int16) - truth)[core].mean()
    err_zero = np.abs(mid_zero.astype(np.int16) - truth)[core].mean()
    assert err_flow < 0.5 * err_zero, (err_flow, err_zero)


@pytest.mark.gpu
@pytest.mark.parametrize("size,levels", [((67, 35), 3), ((130, 17), 4), ((64, 16), 2), ((33, 33), 6), ((5, 3), 3),
                                         ((131, 203), 3), ((249, 130), 2)])  # several strips / row blocks of the streamed kernels
def test_flow_estimate_ragged_sizes(nsc, oracle_mod, size, levels):
    w, h = size
    a, b = oracle_mod.gen_noise(w, h, 41), oracle_mod.gen_noise(w, h, 42)
    fe = nsc.FlowEstimator(levels=levels, coarse_iterations=11, refine_iterations=3)
    want = oracle_mod.flow_estimate(a, b, levels, 11, 3, fe.lambda_)
    for tiled in (1, 2, 3, 0):
        fe.set_tiled(tiled)
        assert np.array_equal(fe.estimate(a, b, w, h), want), (size, levels, tiled)


@pytest.mark.gpu
@pytest.mark.parametrize("levels,coarse,refine", [(1, 9, 0), (1, 0, 0), (3, 0, 4), (3, 5, 0), (2, 17, 9)])
def test_flow_estimate_iteration_edge_cases(nsc, oracle_mod, levels, coarse, refine):
    """One level (the coarse launch writes the caller's buffer), no coarse steps (zero flow has to be
    materialised), no refine steps (the upsampled flow is the result), launch splits of 9 and 17 steps."""
    w, h = 96, 50
    a, b = oracle_mod.gen_noise(w, h, 7), oracle_mod.gen_noise(w, h, 8)
    fe = nsc.FlowEstimator(levels=levels, coarse_iterations=coarse, refine_iterations=refine)
    want = oracle_mod.flow_estimate(a, b, levels, coarse, refine, fe.lambda_)
    for tiled in (1, 2, 3, 0):
        fe.set_tiled(tiled)
        assert np.array_equal(fe.estimate(a, b, w, h), want), (levels, coarse, refine, tiled)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1037, 1029), (613, 517)])
def test_flow_estimate_large_tiles(nsc, oracle_mod, size):
    """Level 0 big enough for the 32-wide tile kernel, ragged in both directions: >= 1024 tiles (256 threads per
    tile) and 256..1023 tiles (1024 threads per tile); level 1 takes the next smaller class each time."""
    w, h = size
    a, b = oracle_mod.gen_noise(w, h, 11), oracle_mod.gen_noise(w, h, 12)
    fe = nsc.FlowEstimator(levels=2, coarse_iterations=4, refine_iterations=5)
    assert np.array_equal(fe.estimate(a, b, w, h), oracle_mod.flow_estimate(a, b, 2, 4, 5, fe.lambda_))


@pytest.mark.gpu
def test_flow_estimate_device_path(nsc, oracle_mod):
    import torch

    w, h = 160, 90
    a, b = _smooth(w, h, 0.0), _smooth(w, h, 1.5)
    dev = torch.device("cuda:0")
    da, db = put(a), put(b)
    dflow = guarded.empty((h, w, 2), dtype=torch.float32, device=dev)
    fe = nsc.FlowEstimator(levels=4, coarse_iterations=30, refine_iterations=8)
    fe.estimate_device(da.data_ptr(), db.data_ptr(), w, h, dflow.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(fetch(dflow), oracle_mod.flow_estimate(a, b, 4, 30, 8, fe.lambda_))
    # and straight into the warp on the device
    it = nsc.WgpuFrameInterpolator()
    out = guarded.empty((h, w, 4), dtype=torch.uint8, device=dev)
    it.interpolate_device(da.data_ptr(), w * h * 4, db.data_ptr(), w * h * 4, dflow.data_ptr(), w, h, 0.5, out.data_ptr(), 1,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(fetch(out), oracle_mod.warp_blend(a, b, fetch(dflow), 0.5))


@pytest.mark.gpu
@pytest.mark.parametrize("n_frames", [2, 3, 5, 67, 103, 153])
def test_flow_estimate_device_stream_equals_pairwise(nsc, oracle_mod, n_frames):
    """The stream entry point takes chunks of up to 150 pairs (100 until round 6) through every stage together (pairs on the grid's z axis,
    each frame's pyramid built once per chunk; 153 frames: two chunks): every flow must still be the oracle's flow of its own pair -- also
    with the shader-shaped kernels, which go pair by pair."""
    import torch

    w, h = 97, 45
    frames = np.stack([oracle_mod.gen_noise(w, h, 60 + k) for k in range(n_frames)])
    dev = torch.device("cuda:0")
    d_frames = put(frames)
    d_flows = guarded.full((n_frames - 1, h, w, 2), float("nan"), dtype=torch.float32, device=dev)
    fe = nsc.FlowEstimator(levels=3, coarse_iterations=9, refine_iterations=3)
    want = [oracle_mod.flow_estimate(frames[k], frames[k + 1], 3, 9, 3, fe.lambda_) for k in range(n_frames - 1)]
    for tiled in (1, 2, 3, 0):
        fe.set_tiled(tiled)
        d_flows.fill_(float("nan"))
        fe.estimate_device_stream(d_frames.data_ptr(), n_frames, w, h, d_flows.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = fetch(d_flows)
        for k in range(n_frames - 1):
            assert np.array_equal(got[k], want[k]), (k, tiled)
    with pytest.raises(Exception):
        fe.estimate_device_stream(d_frames.data_ptr(), 1, w, h, d_flows.data_ptr(), 0)


@pytest.mark.gpu
def test_flow_estimate_device_stream_big_batch(nsc, oracle_mod):
    """A batch whose levels fall into different tile classes (>= 1024 tiles of 32x32 over the batch at level 0:
    32-wide tiles with 256 threads; fewer at the coarser levels: 1024 threads per tile)."""
    import torch

    w, h, n_frames = 333, 262, 13
    frames = np.stack([oracle_mod.gen_noise(w, h, 90 + k) for k in range(n_frames)])
    dev = torch.device("cuda:0")
    d_frames = put(frames)
    d_flows = guarded.full((n_frames - 1, h, w, 2), float("nan"), dtype=torch.float32, device=dev)
    fe = nsc.FlowEstimator(levels=3, coarse_iterations=7, refine_iterations=6)
    fe.estimate_device_stream(d_frames.data_ptr(), n_frames, w, h, d_flows.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = fetch(d_flows)
    for k in range(n_frames - 1):
        assert np.array_equal(got[k], oracle_mod.flow_estimate(frames[k], frames[k + 1], 3, 7, 6, fe.lambda_)), k


@pytest.mark.gpu
def test_pipeline_step_motion_matches_stage_by_stage_oracle(nsc, oracle_mod):
    """FramePipeline.step_motion: per-pair flow -> warp + blend with it -> x2 Lanczos of real and in-between frames,
    device-resident over a small stream; every stage equals the oracle's (bit-exact flow and warp)."""
    import torch

    w, h, n = 160, 96, 3
    frames = np.stack([_smooth(w, h, 1.5 * k) for k in range(n + 1)])
    dev = torch.device("cuda:0")
    d_frames = put(frames)
    pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5, lanczos_mode="exact")
    mid, up_real, up_mid = guarded.like(pipe.alloc(n, dev))
    flows = guarded.empty((n, h, w, 2), dtype=torch.float32, device=dev)
    pipe.step_motion(d_frames, flows, mid, up_real, up_mid, torch.cuda.current_stream().cuda_stream,
                     levels=3, coarse_iterations=20, refine_iterations=5)
    torch.cuda.synchronize()
    lam = pipe._flow.lambda_
    for k in range(n):
        want_flow = oracle_mod.flow_estimate(frames[k], frames[k + 1], 3, 20, 5, lam)
        assert np.array_equal(fetch(flows[k]), want_flow), k
        want_mid = oracle_mod.warp_blend(frames[k], frames[k + 1], want_flow, 0.5)
        assert np.array_equal(fetch(mid[k]), want_mid), k
        assert np.array_equal(fetch(up_mid[k]), oracle_mod.lanczos3(want_mid, 2 * w, 2 * h)), k
        assert np.array_equal(fetch(up_real[k]), oracle_mod.lanczos3(frames[k], 2 * w, 2 * h)), k
    # the same step as a two-stream pipeline over chunks (the estimator of chunk i+1 beside warp + upscales of chunk i): same bytes
    want = [t.clone() for t in (flows, mid, up_real, up_mid)]
    for chunk in (1, 2):
        for t in (flows, mid, up_real, up_mid):
            t.zero_()
        pipe.step_motion(d_frames, flows, mid, up_real, up_mid, torch.cuda.current_stream().cuda_stream,
                         levels=3, coarse_iterations=20, refine_iterations=5, pipelined=True, chunk=chunk)
        torch.cuda.synchronize()
        for got, w_ in zip((flows, mid, up_real, up_mid), want):
            assert torch.equal(got, w_), chunk


# ---- FAST mode of the estimators (nus_flow_set_mode): tolerance against the exact oracle ---------------------------------

def _flow_close(got, want, tol=1e-3):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return float(d.max()) <= tol, float(d.max())


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,levels,coarse,refine", [(97, 45, 3, 9, 3), (333, 262, 3, 7, 6), (160, 90, 4, 30, 8), (613, 517, 2, 11, 5),
                                                       (64, 64, 1, 13, 0), (129, 70, 2, 1, 1),
                                                       # levels narrower than a 16-byte load, single columns / rows, more than four strips
                                                       (13, 9, 3, 5, 2), (9, 7, 3, 4, 2), (5, 5, 2, 3, 1), (3, 33, 2, 3, 2), (1030, 5, 2, 2, 2)])
def test_flow_fast_mode_within_a_thousandth_of_a_pixel(nsc, oracle_mod, w, h, levels, coarse, refine):
    """FAST mode (separable 3x3 sums, multiply by 1/9, precomputed reciprocal, FMAs; every level in the register-pipelined
    kernel): the flow of every estimator entry point within 1e-3 px of the exact oracle's, on smooth moving content and on
    noise (large gradients, chaotic flow), ragged sizes, step counts that split into launches of 1..5 steps, several pairs."""
    import torch

    dev = torch.device("cuda:0")
    for kind in ("smooth", "noise"):
        n_frames = 4
        frames = np.stack([_smooth(w, h, 1.3 * k) if kind == "smooth" else oracle_mod.gen_noise(w, h, 200 + k) for k in range(n_frames)])
        fe = nsc.FlowEstimator(levels=levels, coarse_iterations=coarse, refine_iterations=refine)
        assert fe.mode == "exact"
        fe.set_mode("fast")
        assert fe.mode == "fast"
        want = [oracle_mod.flow_estimate(frames[k], frames[k + 1], levels, coarse, refine, fe.lambda_) for k in range(n_frames - 1)]
        d_frames = put(frames)
        d_flows = guarded.full((n_frames - 1, h, w, 2), float("nan"), dtype=torch.float32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        # kernels by size (1): a batch this small takes the exact LDS-tile kernels, whose result meets the contract trivially;
        # streamed kernel forced (3): the FAST kernels themselves -- k_pyramid_fast, k_hs_stream_fast -- at these ragged sizes
        for tiled in (3, 1):
            fe.set_tiled(tiled)
            d_flows.fill_(float("nan"))
            fe.estimate_device_stream(d_frames.data_ptr(), n_frames, w, h, d_flows.data_ptr(), s)
            torch.cuda.synchronize()
            got = fetch(d_flows)
            assert np.isfinite(got).all()
            for k in range(n_frames - 1):
                ok, mx = _flow_close(got[k], want[k])
                assert ok, (kind, "stream", tiled, k, mx)
            if tiled == 3:
                assert any(not np.array_equal(got[k], want[k]) for k in range(n_frames - 1)), "the FAST kernels did not run"
            one = guarded.full((h, w, 2), float("nan"), dtype=torch.float32, device=dev)
            fe.estimate_device(d_frames[1].data_ptr(), d_frames[2].data_ptr(), w, h, one.data_ptr(), s)
            torch.cuda.synchronize()
            assert np.array_equal(fetch(one), got[1]), (kind, tiled, "a pair alone = the same pair inside a stream")
            assert np.array_equal(fe.estimate(frames[0], frames[1], w, h), got[0]), (kind, tiled, "host entry point")
        fe.set_mode("exact")  # and back: bit-exact again
        fe.set_tiled(1)
        assert np.array_equal(fe.estimate(frames[0], frames[1], w, h), want[0])
    with pytest.raises(ValueError):
        fe.set_mode("sloppy")


@pytest.mark.gpu
def test_flow_fast_mode_1080p_contract(nsc, oracle_mod):
    """The contract of nus_flow_set_mode(NUS_FLOW_FAST) at the size it is for: 1080p, 3 levels, 50 + 10 + 10 steps -- flow within
    1e-3 px of orc_flow_estimate, and the frame interpolated with it within 1 LSB of the frame interpolated with the exact
    flow, on fewer than 0.1 % of the samples."""
    import torch

    w, h = 1920, 1080
    dev = torch.device("cuda:0")
    a, b = _smooth(w, h, 0.0), _smooth(w, h, 1.75)
    rng = np.random.default_rng(7)
    for img in (a, b):  # sensor-like noise on top: the flow is not a clean constant
        img[..., :3] = np.clip(img[..., :3].astype(np.int16) + rng.integers(-3, 4, size=(h, w, 3)), 0, 255).astype(np.uint8)
    fe = nsc.FlowEstimator(levels=3, coarse_iterations=50, refine_iterations=10)
    want = oracle_mod.flow_estimate(a, b, 3, 50, 10, fe.lambda_)
    fe.set_mode("fast")
    frames = put(np.stack([a, b]))
    flow = guarded.empty((1, h, w, 2), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    fe.estimate_device_stream(frames.data_ptr(), 2, w, h, flow.data_ptr(), s)
    torch.cuda.synchronize()
    got = fetch(flow[0])
    ok, mx = _flow_close(got, want)
    assert ok, mx
    it = nsc.WgpuFrameInterpolator()
    out = guarded.empty((h, w, 4), dtype=torch.uint8, device=dev)
    fb = w * h * 4
    it.interpolate_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, flow.data_ptr(), w, h, 0.5, out.data_ptr(), 1, s)
    torch.cuda.synchronize()
    ref = oracle_mod.warp_blend(a, b, want, 0.5)
    d = np.abs(fetch(out).astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (int(d.max()), float((d > 0).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,levels,coarse,refine", [(333, 262, 3, 7, 6), (613, 517, 2, 11, 5), (97, 45, 3, 9, 3), (1030, 5, 2, 2, 2),
                                                       (5, 5, 2, 3, 1), (1920, 1080, 3, 10, 5), (129, 700, 2, 4, 3), (160, 90, 2, 30, 8), (480, 270, 1, 50, 0),
                                                       (333, 100, 1, 37, 0)])
def test_flow_fast_ring_form_equals_the_shifting_form(nsc, oracle_mod, w, h, levels, coarse, refine, monkeypatch):
    """Round 5: k_hs_stream_fast keeps every queue of its pipeline in rings with compile-time slot indices (six copies of the pass,
    no register moves); the shifting form of rounds 3-4 is still in the library (launches of more than 5 steps use it) and
    NUS_HS_FAST_SHIFT=1 selects it everywhere.  Same operations on the same operands in the same order: the flows of the two
    forms are bit-identical -- ragged sizes, blocks shorter than a turn of the rings, several row blocks, 1..5 steps per launch."""
    import torch

    dev = torch.device("cuda:0")
    n_frames = 3
    frames = np.stack([oracle_mod.gen_noise(w, h, 900 + k) if (w * h) % 2 else _smooth(w, h, 1.1 * k) for k in range(n_frames)])
    d_frames = put(frames)
    s = torch.cuda.current_stream().cuda_stream
    got = {}
    for form in ("ring", "shift"):
        if form == "shift":
            monkeypatch.setenv("NUS_HS_FAST_SHIFT", "1")
        else:
            monkeypatch.delenv("NUS_HS_FAST_SHIFT", raising=False)
        fe = nsc.FlowEstimator(levels=levels, coarse_iterations=coarse, refine_iterations=refine)
        fe.set_mode("fast")
        fe.set_tiled(3)  # the streamed kernels whatever the batch size
        flows = guarded.full((n_frames - 1, h, w, 2), float("nan"), dtype=torch.float32, device=dev)
        fe.estimate_device_stream(d_frames.data_ptr(), n_frames, w, h, flows.data_ptr(), s)
        torch.cuda.synchronize()
        got[form] = fetch(flows)
        assert np.isfinite(got[form]).all()
    monkeypatch.delenv("NUS_HS_FAST_SHIFT", raising=False)
    assert np.array_equal(got["ring"], got["shift"])


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,levels,coarse,refine", [(160, 96, 3, 20, 10), (333, 262, 3, 7, 7), (97, 45, 2, 9, 5), (64, 64, 1, 13, 0),
                                                       (130, 70, 2, 4, 0), (1030, 6, 2, 2, 2), (2, 2, 1, 3, 0), (960, 540, 3, 10, 10)])
@pytest.mark.parametrize("in_kernel", [False, True])
def test_interpolate_device_stream_equals_estimate_then_warp(nsc, oracle_mod, w, h, levels, coarse, refine, in_kernel, monkeypatch):
    """nus_flow_interpolate_device_stream (round 5: pyramid -> flow -> warp as ONE pipeline, wgpu_interpolator.rs:881-935): in every
    kernel mode -- EXACT by size, EXACT pair by pair, FAST with the streamed kernels forced (the last Horn-Schunck launch of the finest
    level warps with the flow it has just finished; also with 7 = 4 + 3 steps, with no refinement step at all, with one level) -- the
    in-between frames are bit for bit those of nus_flow_estimate_device_stream followed by the FMA-mode warp kernel, with the flows
    stored or not; the flows, when asked for, are the estimator's; and in EXACT mode everything equals the oracle within the warp's
    FMA contract (<= 1 LSB)."""
    import torch

    # in_kernel: NUS_HS_FUSED_WARP=1 -- the finest level's last FAST launch warps with the flow it has just finished (HsWarp)
    if in_kernel:
        monkeypatch.setenv("NUS_HS_FUSED_WARP", "1")
    else:
        monkeypatch.delenv("NUS_HS_FUSED_WARP", raising=False)
    dev = torch.device("cuda:0")
    n_frames, t = 4, 0.3
    frames = np.stack([_smooth(w, h, 1.2 * k) if k % 2 else oracle_mod.gen_noise(w, h, 50 + k) // 2 + _smooth(w, h, 0.7 * k) // 2
                       for k in range(n_frames)])
    d_frames = put(frames)
    s = torch.cuda.current_stream().cuda_stream
    fb = w * h * 4
    it = nsc.WgpuFrameInterpolator()
    it.set_mode("fma")
    for mode, tiled in (("fast", 3), ("fast", 1), ("exact", 1), ("exact", 0)):
        fe = nsc.FlowEstimator(levels=levels, coarse_iterations=coarse, refine_iterations=refine)
        fe.set_mode(mode)
        fe.set_tiled(tiled)
        flows = guarded.full((n_frames - 1, h, w, 2), float("nan"), dtype=torch.float32, device=dev)
        fe.estimate_device_stream(d_frames.data_ptr(), n_frames, w, h, flows.data_ptr(), s)
        want_mid = guarded.zeros((n_frames - 1, h, w, 4), dtype=torch.uint8, device=dev)
        it.interpolate_device(d_frames.data_ptr(), fb, d_frames.data_ptr() + fb, fb, flows.data_ptr(), w, h, t, want_mid.data_ptr(),
                              n_frames - 1, s)
        torch.cuda.synchronize()
        for with_flows in (True, False):
            got_flows = torch.full_like(flows, float("nan"))
            got_mid = torch.full_like(want_mid, 0xAB)
            fe.interpolate_device_stream(d_frames.data_ptr(), n_frames, w, h, t, got_mid.data_ptr(),
                                         got_flows.data_ptr() if with_flows else 0, s)
            torch.cuda.synchronize()
            assert torch.equal(got_mid, want_mid), (mode, tiled, with_flows)
            if with_flows:
                assert torch.equal(got_flows, flows), (mode, tiled)
            else:
                assert bool(torch.isnan(got_flows).all())
        if mode == "exact" and tiled == 1 and not in_kernel:
            for k in range(n_frames - 1):
                fl = oracle_mod.flow_estimate(frames[k], frames[k + 1], levels, coarse, refine, fe.lambda_)
                assert np.array_equal(fetch(flows[k]), fl)
                d = np.abs(fetch(want_mid[k]).astype(np.int16) - oracle_mod.warp_blend(frames[k], frames[k + 1], fl, t).astype(np.int16))
                assert d.max() <= 1
    fe = nsc.FlowEstimator(levels=2, coarse_iterations=2, refine_iterations=2)
    with pytest.raises(RuntimeError, match=r"t must be in \[0, 1\]"):
        fe.interpolate_device_stream(d_frames.data_ptr(), n_frames, w, h, 1.5, want_mid.data_ptr(), 0, s)
    with pytest.raises(RuntimeError, match="null device pointer"):
        fe.interpolate_device_stream(d_frames.data_ptr(), n_frames, w, h, 0.5, 0, 0, s)


@pytest.mark.gpu
@pytest.mark.parametrize("in_kernel", [False, True])
def test_step_motion_fused_warp_equals_separate_stages(nsc, oracle_mod, in_kernel, monkeypatch):
    """FramePipeline.step_motion(fused_warp=True), stage after stage and as the two-stream pipeline, with the flows stored and not:
    the in-between frames and both 4K outputs bit-identical to the step with the separate FMA-mode warp."""
    import torch

    if in_kernel:
        monkeypatch.setenv("NUS_HS_FUSED_WARP", "1")
    else:
        monkeypatch.delenv("NUS_HS_FUSED_WARP", raising=False)
    w, h, n = 480, 270, 7
    dev = torch.device("cuda:0")
    frames = put(np.stack([_smooth(w, h, 1.1 * k) for k in range(n + 1)]))
    pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
    pipe.interp.set_mode("fma")
    s = torch.cuda.current_stream().cuda_stream
    want = guarded.like(pipe.alloc(n, dev))
    flows = guarded.empty((n, h, w, 2), dtype=torch.float32, device=dev)
    pipe.step_motion(frames, flows, *want, s, flow_mode="fast")
    torch.cuda.synchronize()
    for kw in (dict(fused_warp=True), dict(fused_warp=True, pipelined=True, chunk=3), dict(fused_warp=True, pipelined=True, chunk=3, no_flows=True),
               dict(fused_warp=True, no_flows=True)):
        no_flows = kw.pop("no_flows", False)
        got = guarded.like(pipe.alloc(n, dev))
        for t_ in got:
            t_.zero_()
        got_flows = None if no_flows else guarded.zeros_like(flows)
        # (with the flows kept in the workspace the hand-off defaults to Rg16Float in FAST mode: "f32" asks for the old bytes)
        pipe.step_motion(frames, got_flows, *got, s, flow_mode="fast", flow_format="f32", **kw)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b), kw
        if got_flows is not None:
            assert torch.equal(got_flows, flows), kw
        if no_flows:
            # the default: the flow handed over as Rg16Float (wgpu_interpolator.rs:275-276) -- the real frames' upscale is untouched,
            # the in-between frame within one count of the f32 hand-off's on few samples, and its upscale follows it
            half = guarded.like(pipe.alloc(n, dev))
            pipe.step_motion(frames, None, *half, s, flow_mode="fast", **kw)
            torch.cuda.synchronize()
            assert torch.equal(half[1], want[1]), kw
            d = (half[0].to(torch.int16) - want[0].to(torch.int16)).abs()
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3, (kw, int(d.max()), float((d > 0).float().mean()))
            d = (half[2].to(torch.int16) - want[2].to(torch.int16)).abs()
            assert int(d.max()) <= 2 and float((d > 0).float().mean()) < 2e-3, (kw, int(d.max()), float((d > 0).float().mean()))
    with pytest.raises(ValueError, match="fused_warp"):
        pipe.step_motion(frames, None, *want, s)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,levels,coarse,refine", [(160, 96, 3, 20, 10), (333, 262, 3, 7, 7), (64, 64, 1, 13, 0), (130, 70, 2, 4, 0),
                                                       (960, 540, 3, 10, 10)])
def test_interpolate_device_stream_rg16float_handoff(nsc, oracle_mod, w, h, levels, coarse, refine):
    """flow_format NUS_FLOW_F16: the flows between estimator and warp as Rg16Float, the reference's live layout
    (wgpu_interpolator.rs:276).  In every kernel mode the f16 flows are the f32 flows of the same estimator rounded to nearest even
    (bit for bit: torch's float16 cast is the same rounding), stored at d_flows or not; the in-between frames are bit for bit the
    FMA-mode warp kernel's on those f16 flows; and against the f32 hand-off they differ by at most 1 LSB."""
    import torch

    dev = torch.device("cuda:0")
    n_frames, t = 4, 0.5
    frames = np.stack([_smooth(w, h, 0.9 * k) for k in range(n_frames)])
    d_frames = put(frames)
    s = torch.cuda.current_stream().cuda_stream
    fb = w * h * 4
    it = nsc.WgpuFrameInterpolator()
    it.set_mode("fma")
    for mode, tiled in (("fast", 3), ("fast", 1), ("exact", 1), ("exact", 0)):
        fe = nsc.FlowEstimator(levels=levels, coarse_iterations=coarse, refine_iterations=refine)
        fe.set_mode(mode)
        fe.set_tiled(tiled)
        flows32 = guarded.empty((n_frames - 1, h, w, 2), dtype=torch.float32, device=dev)
        mid32 = guarded.empty((n_frames - 1, h, w, 4), dtype=torch.uint8, device=dev)
        fe.interpolate_device_stream(d_frames.data_ptr(), n_frames, w, h, t, mid32.data_ptr(), flows32.data_ptr(), s)
        want16 = flows32.to(torch.float16)
        it.set_flow_format("f16")
        want_mid = guarded.zeros_like(mid32)
        it.interpolate_device(d_frames.data_ptr(), fb, d_frames.data_ptr() + fb, fb, want16.data_ptr(), w, h, t, want_mid.data_ptr(),
                              n_frames - 1, s)
        it.set_flow_format("f32")
        torch.cuda.synchronize()
        for with_flows in (True, False):
            # the f16 flows take 4 bytes per cell: a guard of the same size again behind them must stay untouched (a kernel that
            # stored 2 x f32 per cell into the caller's buffer would run over it -- the first version of this path did, round 5)
            cells = (n_frames - 1) * h * w
            buf = guarded.full((2 * cells, 2), 1234.0, dtype=torch.float16, device=dev)
            got16 = buf[:cells].view(n_frames - 1, h, w, 2)
            got16.fill_(float("nan"))
            got_mid = torch.full_like(mid32, 0xAB)
            fe.interpolate_device_stream(d_frames.data_ptr(), n_frames, w, h, t, got_mid.data_ptr(), got16.data_ptr() if with_flows else 0,
                                         s, "f16")
            torch.cuda.synchronize()
            assert bool((buf[cells:] == 1234.0).all()), (mode, tiled, with_flows, "wrote past the f16 flows")
            assert torch.equal(got_mid, want_mid), (mode, tiled, with_flows)
            if with_flows:
                assert torch.equal(got16.view(torch.int16), want16.view(torch.int16)), (mode, tiled)
            else:
                assert bool(torch.isnan(got16).all())
        d = (want_mid.to(torch.int16) - mid32.to(torch.int16)).abs()
        assert int(d.max()) <= 1, (mode, tiled)
    with pytest.raises(ValueError):
        fe.interpolate_device_stream(d_frames.data_ptr(), n_frames, w, h, t, mid32.data_ptr(), 0, s, "bf16")
