"""The one-launch pipeline step (nus_upscaler_upscale_unit_device, k_lanczos3_x2<.., UNIT>): upscale(A), the zero-flow
in-between frame blend(A, B) and upscale(blend(A, B)) for a batch of pairs from a single launch of the x2 resize kernel.
Checked against the three separate stages (bit for bit, every mode) and against the CPU oracle."""
import numpy as np
import pytest
from conftest import guarded  # device outputs between poisoned guard bands (tests/conftest.py)
from nu_scaler_amd.transfer import to_device as put, to_numpy as fetch  # host <-> HBM through nus_upload / nus_download, never
# torch's pageable copies (docs/d2h_fault_analysis.md)

pytestmark = pytest.mark.gpu


def _three_stage(nsc, torch, u, frames, w, h, n, t):
    dev = frames.device
    fb = w * h * 4
    s = torch.cuda.current_stream().cuda_stream
    it = nsc.WgpuFrameInterpolator()
    mid = guarded.empty((n, h, w, 4), dtype=torch.uint8, device=dev)
    up_real = guarded.zeros((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
    up_mid = guarded.zeros_like(up_real)
    it.interpolate_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, 0, w, h, t, mid.data_ptr(), n, s)
    u.upscale_device(frames.data_ptr(), up_real.data_ptr(), n, s)
    u.upscale_device(mid.data_ptr(), up_mid.data_ptr(), n, s)
    torch.cuda.synchronize()
    return mid, up_real, up_mid


@pytest.mark.parametrize("w,h,th", [(128, 40, 0), (496, 50, 12), (240, 37, 6), (964, 45, 7)])
@pytest.mark.parametrize("t", [0.5, 0.3])
def test_unit_step_equals_the_three_stages(nsc, oracle_mod, w, h, th, t):
    """Several strips (240 columns each) and row blocks, ragged last strip / last row block, both wave orders, both
    Lanczos modes, opaque and 4-channel content: every output buffer bit-identical to the three-stage path."""
    import torch

    n = 5
    dev = torch.device("cuda:0")
    fb = w * h * 4
    s = torch.cuda.current_stream().cuda_stream
    for content in ("noise", "opaque", "flat"):
        frames_np = np.stack([oracle_mod.gen_noise(w, h, 700 + i) for i in range(n + 1)])
        if content == "opaque":
            frames_np[..., 3] = 255
            frames_np[2, h // 2:, :, 3] = 17  # one frame with real alpha in its lower half: mixed paths inside one launch
        if content == "flat":  # round 5: any flat alpha takes the 3-channel path -- regions of 0 / 128 / 255 shared by all frames
            frames_np[:, :h // 3, :, 3] = 0   # (the blend of two rows of one alpha has that alpha), and one frame out of step
            frames_np[:, h // 3:2 * h // 3, :, 3] = 128
            frames_np[:, 2 * h // 3:, :, 3] = 255
            frames_np[3, :, :, 3] = 77
        frames = put(frames_np)
        for mode in ("fma", "exact"):
            for order in (1, 0):
                u = nsc.PyWgpuUpscaler("quality", "lanczos3", lanczos_mode=mode)
                if th:
                    u.set_option("rows_per_wave", th)
                u.set_option("unit_order", order)
                u.initialize(w, h, 2 * w, 2 * h)
                assert u.kernel_variant == "lanczos3_x2_regwin"
                want_mid, want_real, want_up_mid = _three_stage(nsc, torch, u, frames, w, h, n, t)
                mid = guarded.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
                up_real = guarded.zeros((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
                up_mid = guarded.zeros_like(up_real)
                u.upscale_unit_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, t, mid.data_ptr(), up_real.data_ptr(),
                                      up_mid.data_ptr(), n, s)
                torch.cuda.synchronize()
                tag = (content, mode, order, w, h, th, t)
                assert torch.equal(mid, want_mid), ("mid", tag)
                assert torch.equal(up_real, want_real), ("up_real", tag)
                assert torch.equal(up_mid, want_up_mid), ("up_mid", tag)
        # the in-between frames against the oracle itself (bit-exact: interpolation/mod.rs:407-411 truncation)
        for i in (0, n - 1):
            assert np.array_equal(fetch(mid[i]), oracle_mod.warp_blend(frames_np[i], frames_np[i + 1], None, t))


def test_unit_step_without_mid_buffer_and_separate_pair_buffers(nsc, oracle_mod):
    """d_mid = NULL: only the two 4K outputs; A and B from two separate buffers with stride 0 (tightly packed)."""
    import torch

    w, h, n = 256, 48, 3
    dev = torch.device("cuda:0")
    a_np = np.stack([oracle_mod.gen_noise(w, h, 40 + i) for i in range(n)])
    b_np = np.stack([oracle_mod.gen_noise(w, h, 90 + i) for i in range(n)])
    a, b = put(a_np), put(b_np)
    u = nsc.PyWgpuUpscaler("quality", "lanczos3", lanczos_mode="exact")
    u.initialize(w, h, 2 * w, 2 * h)
    up_real = guarded.zeros((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
    up_mid = guarded.zeros_like(up_real)
    u.upscale_unit_device(a.data_ptr(), 0, b.data_ptr(), 0, 0.5, 0, up_real.data_ptr(), up_mid.data_ptr(), n,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(n):
        assert np.array_equal(fetch(up_real[i]), oracle_mod.lanczos3(a_np[i], 2 * w, 2 * h))
        m = oracle_mod.warp_blend(a_np[i], b_np[i], None, 0.5)
        assert np.array_equal(fetch(up_mid[i]), oracle_mod.lanczos3(m, 2 * w, 2 * h))


def test_unit_step_bgra_input_and_other_filters(nsc, oracle_mod):
    """The channel swizzle on load and the other x2 filters go through the same kernel."""
    import torch

    w, h, n = 248, 33, 2
    dev = torch.device("cuda:0")
    frames_np = np.stack([oracle_mod.gen_noise(w, h, 5 + i) for i in range(n + 1)])
    frames = put(frames_np)
    fb = w * h * 4
    s = torch.cuda.current_stream().cuda_stream
    for alg, fmt in (("lanczos3", "bgra"), ("bicubic", "rgba"), ("triangle", "rgba")):
        u = nsc.PyWgpuUpscaler("quality", alg)
        u.set_input_format(fmt)
        u.initialize(w, h, 2 * w, 2 * h)
        it = nsc.WgpuFrameInterpolator()
        it.set_input_format(fmt)
        want_mid = guarded.empty((n, h, w, 4), dtype=torch.uint8, device=dev)
        it.interpolate_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, 0, w, h, 0.5, want_mid.data_ptr(), n, s)
        want_real = guarded.zeros((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
        u.upscale_device(frames.data_ptr(), want_real.data_ptr(), n, s)
        u.set_input_format("rgba")  # the in-between frames are RGBA already
        want_up_mid = guarded.zeros_like(want_real)
        u.upscale_device(want_mid.data_ptr(), want_up_mid.data_ptr(), n, s)
        u.set_input_format(fmt)
        mid = guarded.zeros_like(want_mid)
        up_real, up_mid = guarded.zeros_like(want_real), guarded.zeros_like(want_real)
        u.upscale_unit_device(frames.data_ptr(), fb, frames.data_ptr() + fb, fb, 0.5, mid.data_ptr(), up_real.data_ptr(),
                              up_mid.data_ptr(), n, s)
        torch.cuda.synchronize()
        assert torch.equal(mid, want_mid) and torch.equal(up_real, want_real) and torch.equal(up_mid, want_up_mid), alg


@pytest.mark.parametrize("rows_per_wave", [108, 36])
def test_unit_step_1080p_bench_shape(nsc, oracle_mod, rows_per_wave):
    """The launch shape bench.py times (1080p -> 4K, 108 rows per wave for its 300-unit batches, 36 for smaller ones, a sliding
    stream) on the opaque gradient and on noise: all three outputs against the oracle (Lanczos FMA mode: <= 1 LSB, < 0.1 % of the
    samples different)."""
    import torch

    w, h, n = 1920, 1080, 6
    dev = torch.device("cuda:0")
    for pattern in ("gradient", "noise"):
        gen = (lambda k: oracle_mod.gen_gradient(w, h, k)) if pattern == "gradient" else (lambda k: oracle_mod.gen_noise(w, h, 60 + k))
        frames_np = np.stack([gen(k) for k in range(n + 1)])
        frames = put(frames_np)
        pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
        pipe.upscaler.set_option("rows_per_wave", rows_per_wave)
        mid, up_real, up_mid = guarded.like(pipe.alloc(n, dev))
        for tns in (mid, up_real, up_mid):
            tns.zero_()
        pipe.step_unit(frames, mid, up_real, up_mid, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for k in (0, n - 1):
            m = oracle_mod.warp_blend(frames_np[k], frames_np[k + 1], None, 0.5, threads=0)
            assert np.array_equal(fetch(mid[k]), m), (pattern, k)
            for got, src in ((up_real, frames_np[k]), (up_mid, m)):
                want = oracle_mod.lanczos3(src, 2 * w, 2 * h, threads=0).astype(np.int16)
                d = np.abs(fetch(got[k]).astype(np.int16) - want)
                assert d.max() <= 1 and (d > 0).mean() < 1e-3, (pattern, k, int(d.max()), float((d > 0).mean()))


def test_unit_step_errors(nsc):
    import torch

    dev = torch.device("cuda:0")
    x = guarded.zeros(1 << 20, dtype=torch.uint8, device=dev)
    u = nsc.PyWgpuUpscaler("quality", "bilinear")
    u.initialize(64, 32, 128, 64)
    with pytest.raises(RuntimeError, match="only the exact-x2 resize kernels"):
        u.upscale_unit_device(x.data_ptr(), 0, x.data_ptr(), 0, 0.5, 0, x.data_ptr(), x.data_ptr(), 1, 0)
    u = nsc.PyWgpuUpscaler("quality", "lanczos3")
    with pytest.raises(RuntimeError, match="not initialized"):
        u.upscale_unit_device(x.data_ptr(), 0, x.data_ptr(), 0, 0.5, 0, x.data_ptr(), x.data_ptr(), 1, 0)
    u.initialize(64, 32, 128, 64)
    with pytest.raises(RuntimeError, match="null device pointer"):
        u.upscale_unit_device(x.data_ptr(), 0, 0, 0, 0.5, 0, x.data_ptr(), x.data_ptr(), 1, 0)
    with pytest.raises(RuntimeError, match="multiples of 16"):
        u.upscale_unit_device(x.data_ptr() + 4, 0, x.data_ptr(), 0, 0.5, 0, x.data_ptr(), x.data_ptr(), 1, 0)
    with pytest.raises(RuntimeError, match=r"t must be in \[0, 1\]"):
        u.upscale_unit_device(x.data_ptr(), 0, x.data_ptr(), 0, 1.5, 0, x.data_ptr(), x.data_ptr(), 1, 0)


def test_unit_step_strided_source_frames_keep_mid_packed(nsc, oracle_mod):
    """a_stride = b_stride = 2 frames (A_k and B_k interleaved in one pool): the in-between frames still land tightly packed
    in d_mid (nuscaler_hip.h), nothing is written behind frame n-1, and all outputs equal the packed call's."""
    import torch

    w, h, n = 496, 40, 4
    dev = torch.device("cuda:0")
    fb = w * h * 4
    s = torch.cuda.current_stream().cuda_stream
    pool_np = np.stack([oracle_mod.gen_noise(w, h, 300 + i) for i in range(2 * n)])
    pool = put(pool_np)
    a, b = pool[0::2].contiguous(), pool[1::2].contiguous()
    u = nsc.PyWgpuUpscaler("quality", "lanczos3")
    u.set_option("rows_per_wave", 12)
    u.initialize(w, h, 2 * w, 2 * h)
    want_mid = guarded.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
    want_real = guarded.zeros((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
    want_up_mid = guarded.zeros_like(want_real)
    u.upscale_unit_device(a.data_ptr(), 0, b.data_ptr(), 0, 0.5, want_mid.data_ptr(), want_real.data_ptr(), want_up_mid.data_ptr(), n, s)
    mid = torch.full((2 * n, h, w, 4), 0xAB, dtype=torch.uint8, device=dev)  # n frames + n guard frames
    up_real, up_mid = guarded.zeros_like(want_real), guarded.zeros_like(want_real)
    u.upscale_unit_device(pool.data_ptr(), 2 * fb, pool.data_ptr() + fb, 2 * fb, 0.5, mid.data_ptr(), up_real.data_ptr(),
                          up_mid.data_ptr(), n, s)
    torch.cuda.synchronize()
    assert torch.equal(mid[:n], want_mid)
    assert bool((mid[n:] == 0xAB).all()), "in-between frames written at the source stride"
    assert torch.equal(up_real, want_real) and torch.equal(up_mid, want_up_mid)
    for i in range(n):
        assert np.array_equal(fetch(mid[i]), oracle_mod.warp_blend(pool_np[2 * i], pool_np[2 * i + 1], None, 0.5))


def test_unit_step_64_units_1080p_equals_three_stages_everywhere(nsc, oracle_mod):
    """A batch big enough for everything a 300-unit step switches on (the edge passes beside the main kernel, rows per wave chosen
    by the library) at full size: all three outputs of all 64 units bit-identical to the three separate stages on both contents,
    and -- the size-independent property -- to themselves across two wave orders and a second run."""
    import torch

    from nu_scaler_amd import synthetic as syn

    w, h, n = 1920, 1080, 64
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    for pattern in ("gradient", "noise"):
        frames = (syn.gradient_stream_torch if pattern == "gradient" else syn.noise_stream_torch)(n + 1, w, h, dev)
        pipe = nsc.FramePipeline(w, h, 2, "lanczos3", 0.5)
        want = guarded.like(pipe.alloc(n, dev))
        pipe.step(frames, *want, s)
        torch.cuda.synchronize()
        got = guarded.like(pipe.alloc(n, dev))
        for order in (1, 0, 1):
            pipe.upscaler.set_option("unit_order", order)
            for t_ in got:
                t_.zero_()
            pipe.step_unit(frames, *got, s)
            torch.cuda.synchronize()
            for name, a, b in zip(("mid", "up_real", "up_mid"), got, want):
                assert torch.equal(a, b), (pattern, order, name)
        # and one unit against the oracle, edge columns included
        k = n - 1
        m = oracle_mod.warp_blend(fetch(frames[k]), fetch(frames[k + 1]), None, 0.5, threads=0)
        assert np.array_equal(fetch(got[0][k]), m)
        d = np.abs(fetch(got[2][k]).astype(np.int16) - oracle_mod.lanczos3(m, 2 * w, 2 * h, threads=0).astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
        del want, got, frames
