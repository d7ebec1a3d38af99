"""FSR1-style EASU + RCAS ("next" row).

CPU: the C oracle against its independent numpy twin (PARITY UNPINNED: the reference keeps the two
WGSL shaders in nu_scaler_core/src/upscale/fsr.rs:24-260 but never dispatches them, so there is no
fixture to pin either restatement to).  GPU: the HIP kernels, through the C ABI, bit-exact against
the C oracle -- each pass alone and the fused pair.
"""
import numpy as np
import pytest
from conftest import guarded  # device outputs between poisoned guard bands (tests/conftest.py)
from nu_scaler_amd.transfer import to_device as put, to_numpy as fetch  # host <-> HBM through nus_upload / nus_download, never
# torch's pageable copies (docs/d2h_fault_analysis.md)

SIZES = [((32, 24), (64, 48)), ((17, 13), (40, 29)), ((20, 20), (20, 20)), ((33, 21), (25, 17)), ((1, 1), (3, 2)),
         ((5, 3), (130, 67))]


@pytest.mark.parametrize("dims", SIZES)
def test_oracle_c_equals_numpy_twin(oracle_mod, dims):
    from oracle import oracle_np as onp
    (w, h), (ow, oh) = dims
    for img in (oracle_mod.gen_noise(w, h, 31), oracle_mod.gen_gradient(w, h)):
        for s in (0.0, 0.3):
            e = oracle_mod.fsr_easu(img, ow, oh, s)
            assert np.array_equal(e, onp.fsr_easu(img, ow, oh, s))
            assert np.array_equal(oracle_mod.fsr_rcas(e, 0.7), onp.fsr_rcas(e, 0.7))
            assert np.array_equal(oracle_mod.fsr1(img, ow, oh, s, 0.7), oracle_mod.fsr_rcas(e, 0.7))


def test_oracle_fsr_properties(oracle_mod):
    # flat image: EASU's weights normalise, so it returns the value up to the truncating pack; alpha -> 255
    flat = np.full((9, 11, 4), 77, np.uint8)
    flat[..., 3] = 3
    e = oracle_mod.fsr_easu(flat, 22, 18, 0.0)
    assert set(np.unique(e[..., :3])) <= {76, 77} and (e[..., 3] == 255).all()
    # ... and RCAS's laplacian vanishes there up to f32 rounding (4c - c - c - c - c is not exact)
    assert set(np.unique(oracle_mod.fsr_rcas(flat, 0.9)[..., :3])) <= {76, 77}
    # zero sharpness: RCAS is unpack -> pack, which loses at most one count to the truncation
    img = oracle_mod.gen_noise(16, 8, 5)
    r = oracle_mod.fsr_rcas(img, 0.0)
    d = img[..., :3].astype(int) - r[..., :3].astype(int)
    assert d.min() >= 0 and d.max() <= 1 and (r[..., 3] == 255).all()


def _run(nsc, alg, img, ow, oh, easu=-1.0, rcas=-1.0, quality="quality", fast=False):
    u = nsc.PyWgpuUpscaler(quality, alg)
    if fast:
        u.set_option("fsr_fast", 1)
    u.set_sharpness(easu, rcas)
    ih, iw = img.shape[:2]
    u.initialize(iw, ih, ow, oh)
    return np.frombuffer(u.upscale(img.tobytes()), np.uint8).reshape(oh, ow, 4), u


GPU_SIZES = SIZES + [((64, 36), (128, 72)), ((320, 180), (640, 360)), ((100, 70), (257, 131)), ((70, 40), (64, 33))]


@pytest.mark.gpu
@pytest.mark.parametrize("dims", GPU_SIZES)
def test_gpu_easu_rcas_and_fused_bit_exact(nsc, oracle_mod, dims):
    (w, h), (ow, oh) = dims
    for img in (oracle_mod.gen_noise(w, h, 32), oracle_mod.gen_gradient(w, h)):
        for es in (0.0, 0.3):
            want_e = oracle_mod.fsr_easu(img, ow, oh, es)
            got_e, u = _run(nsc, "easu", img, ow, oh, easu=es)
            assert u.kernel_variant == "fsr1_easu_tile"
            assert np.array_equal(got_e, want_e)
            got_f, u = _run(nsc, "fsr1", img, ow, oh, easu=es, rcas=0.7)
            assert u.kernel_variant == "fsr1_easu_rcas_fused_lds"
            assert np.array_equal(got_f, oracle_mod.fsr1(img, ow, oh, es, 0.7))
        got_r, u = _run(nsc, "rcas", img, w, h, rcas=0.6)
        assert u.kernel_variant == "fsr1_rcas_rows"
        assert np.array_equal(got_r, oracle_mod.fsr_rcas(img, 0.6))


@pytest.mark.gpu
def test_gpu_fsr1_quality_default_and_errors(nsc, oracle_mod):
    img = oracle_mod.gen_noise(48, 27, 33)
    for q, s in (("ultra", 0.8), ("balanced", 0.6)):
        got, _ = _run(nsc, "fsr1", img, 96, 54, quality=q)
        assert np.array_equal(got, oracle_mod.fsr1(img, 96, 54, 0.0, s))
    u = nsc.PyWgpuUpscaler("quality", "rcas")
    with pytest.raises(RuntimeError, match="same-size"):
        u.initialize(8, 8, 16, 16)


@pytest.mark.gpu
def test_gpu_fsr1_1080p_to_4k_device_batch(nsc, oracle_mod):
    """BASELINE-size frames through the device path against the oracle.  Round 5: frames of 1 MiB and more take two launches --
    EASU into scratch images of the handle (4 frames at a time), the row-walking RCAS out of them -- instead of the fused LDS tile;
    7 frames per call = two chunks; a second call on ANOTHER stream right behind the first must not overwrite the scratch images the
    first is still reading (the handle's event); option fsr_two_pass = 0 keeps the fused tile, same bytes."""
    import torch
    w, h = 1920, 1080
    frames = np.stack([oracle_mod.gen_gradient(w, h, k) for k in range(2)] + [oracle_mod.gen_noise(w, h, 34 + k) for k in range(5)])
    n = len(frames)
    u = nsc.PyWgpuUpscaler("quality", "fsr1")
    u.initialize(w, h, 2 * w, 2 * h)
    assert u.kernel_variant == "fsr1_easu_then_rcas_rows"
    d_in = put(frames)
    d_out = guarded.empty((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device="cuda")
    d_out2 = guarded.empty_like(d_out)
    side = torch.cuda.Stream()
    u.upscale_device(d_in.data_ptr(), d_out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    rev = d_in.flip(0).contiguous()
    torch.cuda.synchronize()
    u.upscale_device(d_in.data_ptr(), d_out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    u.upscale_device(rev.data_ptr(), d_out2.data_ptr(), n, side.cuda_stream)  # other frames, other stream, same scratch
    torch.cuda.synchronize()
    got, got2 = fetch(d_out), fetch(d_out2)
    for k in (0, 2, 3, 4, n - 1):
        want = oracle_mod.fsr1(frames[k], 2 * w, 2 * h, 0.0, 0.7)
        assert np.array_equal(got[k], want), k
        assert np.array_equal(got2[n - 1 - k], want), k
    f = nsc.PyWgpuUpscaler("quality", "fsr1")
    f.set_option("fsr_two_pass", 0)
    f.initialize(w, h, 2 * w, 2 * h)
    assert f.kernel_variant == "fsr1_easu_rcas_fused_lds"
    d_out2.zero_()
    f.upscale_device(d_in.data_ptr(), d_out2.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(d_out, d_out2)
    with pytest.raises(RuntimeError, match="before initialize"):
        f.set_option("fsr_two_pass", 1)


FAST_SIZES = [((64, 36), (128, 72)), ((320, 180), (640, 360)), ((100, 70), (257, 131)), ((17, 13), (40, 29)), ((5, 3), (130, 67)),
              ((1, 1), (3, 2)), ((480, 270), (960, 540)), ((32, 24), (64, 48))]


@pytest.mark.gpu
@pytest.mark.parametrize("dims", FAST_SIZES)
def test_gpu_fsr_fast_mode_contract(nsc, oracle_mod, dims):
    """Option "fsr_fast" (round 5): EASU's tap distances and piece decisions as the shader computes them, FsrCubic from a 128-cells-
    per-unit table with linear interpolation, FMA sums, reciprocal + Newton step; direction weights and RCAS unchanged.  Contract:
    EASU within 1 LSB of orc_fsr_easu on every sample (noise, gradient, flat and an edge image whose direction weight is exactly
    0.5 -- taps at distance exactly 2, the discontinuity of the weight); the fused pair bit for bit orc_fsr_rcas of the FAST EASU
    image.  EXACT (the default) stays bit-exact, and the two modes really differ somewhere."""
    (w, h), (ow, oh) = dims
    flat = np.full((h, w, 4), 77, np.uint8)
    edge = np.zeros((h, w, 4), np.uint8)
    edge[:, w // 2:] = 200  # a vertical edge: vgx = 0 on both sides of it, equal gradients in flat areas -> wx = 0.5 exactly
    edge[h // 2:, :] //= 2
    differs = False
    for img in (oracle_mod.gen_noise(w, h, 77), oracle_mod.gen_gradient(w, h), flat, edge):
        for es in (0.0, 0.3):
            want_e = oracle_mod.fsr_easu(img, ow, oh, es)
            got_e, u = _run(nsc, "easu", img, ow, oh, easu=es, fast=True)
            assert u.kernel_variant == "fsr1_easu_tile"
            d = np.abs(got_e.astype(np.int16) - want_e.astype(np.int16))
            assert d.max() <= 1, (int(d.max()), es)
            assert (got_e[..., 3] == 255).all()
            differs = differs or bool(d.any())
            got_f, u = _run(nsc, "fsr1", img, ow, oh, easu=es, rcas=0.7, fast=True)
            assert u.kernel_variant == ("fsr1_easu_then_rcas_rows" if ow * oh * 4 >= 1 << 20 else "fsr1_easu_rcas_fused_lds")
            assert np.array_equal(got_f, oracle_mod.fsr_rcas(got_e, 0.7)), es
        got_x, _ = _run(nsc, "fsr1", img, ow, oh, easu=0.0, rcas=0.7)  # the default mode is untouched
        assert np.array_equal(got_x, oracle_mod.fsr1(img, ow, oh, 0.0, 0.7))
    if w * h >= 64 * 36:
        assert differs, "the FAST kernel did not run"
    u = nsc.PyWgpuUpscaler("quality", "fsr1")
    with pytest.raises(RuntimeError, match="fsr_fast must be 0 or 1"):
        u.set_option("fsr_fast", 2)


@pytest.mark.gpu
def test_gpu_fsr_fast_1080p_to_4k_device_batch(nsc, oracle_mod):
    """The FAST contract at the BASELINE size through the device path (3 frames per launch)."""
    import torch
    w, h = 1920, 1080
    frames = np.stack([oracle_mod.gen_gradient(w, h, 3), oracle_mod.gen_noise(w, h, 35), oracle_mod.gen_noise(w, h, 36)])
    d_in = put(frames)
    outs = {}
    for alg in ("easu", "fsr1"):
        u = nsc.PyWgpuUpscaler("quality", alg)
        u.set_option("fsr_fast", 1)
        u.initialize(w, h, 2 * w, 2 * h)
        d_out = guarded.empty((3, 2 * h, 2 * w, 4), dtype=torch.uint8, device="cuda")
        u.upscale_device(d_in.data_ptr(), d_out.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs[alg] = fetch(d_out)
    for k in (0, 1):
        d = np.abs(outs["easu"][k].astype(np.int16) - oracle_mod.fsr_easu(frames[k], 2 * w, 2 * h, 0.0).astype(np.int16))
        assert d.max() <= 1
        assert np.array_equal(outs["fsr1"][k], oracle_mod.fsr_rcas(outs["easu"][k], 0.7))
