"""CPU: the C-ABI library loads and exports what include/nuscaler_hip.h declares; host
logic (tables, validation, error texts, sharding, generators) without any GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "nuscaler_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nus_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(nsc):
    L = ctypes.CDLL(nsc._capi.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    bound = {n for n, _, _ in nsc._capi.SIGNATURES}
    assert bound == set(declared), f"ctypes binding out of sync: {bound ^ set(declared)}"
    assert L.nus_abi_version() == 1


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nu_scaler_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "nus_oracle" not in text, f


def test_create_and_enums(nsc):
    C = nsc._capi
    L = C.lib()
    assert not L.nus_upscaler_create(8, C.QUALITY_QUALITY)
    assert b"unknown" in L.nus_last_error()
    assert not L.nus_interp_create(9)
    for tech, want in ((C.TECH_WGPU, b"WgpuBilinearUpscaler"), (C.TECH_FSR, b"WgpuNearestUpscaler"),
                       (C.TECH_DLSS, b"WgpuNearestUpscaler"), (C.TECH_NONE, b"WgpuNearestUpscaler"),
                       (C.TECH_FALLBACK, b"WgpuNearestUpscaler")):
        h = L.nus_upscaler_create_for_technology(tech, C.QUALITY_ULTRA)
        assert L.nus_upscaler_name(h) == want
        assert L.nus_upscaler_quality(h) == C.QUALITY_ULTRA
        assert L.nus_upscaler_set_quality(h, C.QUALITY_BALANCED) == 0 and L.nus_upscaler_quality(h) == C.QUALITY_BALANCED
        assert L.nus_upscaler_set_quality(h, 42) == C.ERR_INVALID_ARGUMENT
        L.nus_upscaler_destroy(h)
    assert L.nus_status_string(C.ERR_NO_DEVICE) == b"no HIP device"


def test_fsr1_names_and_sharpness_defaults(nsc):
    # RCAS default per quality: Nu_scale/src/upscale/fsr3.rs:231-236; EASU default 0 (build-defined)
    for q, want in (("ultra", 0.8), ("quality", 0.7), ("balanced", 0.6), ("performance", 0.5)):
        u = nsc.PyWgpuUpscaler(q, "fsr1")
        assert u.name == "HipFsr1Upscaler"
        e, r = u.get_sharpness()
        assert e == 0.0 and abs(r - want) < 1e-7
    u = nsc.PyWgpuUpscaler("quality", "easu")
    assert u.name == "HipFsrEasuUpscaler"
    u.set_sharpness(0.25, 0.5)
    e, r = u.get_sharpness()
    assert (e, r) == (0.25, 0.5)
    u.set_sharpness(-1.0, -1.0)
    assert u.get_sharpness()[0] == 0.0
    with pytest.raises(RuntimeError, match="sharpness"):
        u.set_sharpness(2.0, 0.5)
    assert nsc.PyWgpuUpscaler("quality", "rcas").name == "HipFsrRcasUpscaler"


def test_pyclass_surface_and_defaults(nsc):
    u = nsc.PyWgpuUpscaler()
    assert u.name == "WgpuNearestUpscaler" and u.upscale_scale == 2.0
    assert nsc.PyWgpuUpscaler("ultra", "bilinear").name == "WgpuBilinearUpscaler"
    assert nsc.PyWgpuUpscaler("nonsense", "nonsense").name == "WgpuNearestUpscaler"  # silent defaults
    assert nsc.PyWgpuUpscaler("quality", "LANCZOS3").name == "HipLanczos3Upscaler"
    with pytest.raises(ValueError, match="Scale factor must be between 1.0 and 4.0"):
        u.upscale_scale = 4.5
    u.upscale_scale = 3.0
    assert u.upscale_scale == 3.0
    for noop in (lambda: u.reload_shader("x.wgsl"), lambda: u.set_thread_count(4),
                 lambda: u.set_buffer_pool_size(3), lambda: u.set_gpu_allocator("balanced")):
        assert noop() is None
    adv = nsc.create_advanced_upscaler("balanced")
    assert adv.name == "WgpuBilinearUpscaler" and adv.get_quality_str() == "balanced" and adv.adaptive_quality
    with pytest.raises(NotImplementedError):
        nsc.create_fsr_upscaler("quality")
    for k in ("QUALITY_ULTRA", "TECH_WGPU", "VENDOR_AMD"):
        assert hasattr(nsc, k)


def test_not_initialized_error_text(nsc):
    u = nsc.PyWgpuUpscaler("quality", "bilinear")
    with pytest.raises(RuntimeError, match=r"Upscaler not initialized\. Call initialize\(\) first\."):
        u.upscale(b"\0" * 16)
    with pytest.raises(RuntimeError, match="not initialized"):
        u.upscale_batch([b"\0" * 16])
    with pytest.raises(RuntimeError, match="not initialized"):
        u.upscale_batch([])
    assert u.get_last_gpu_duration_ms() is None and u.input_size == 0


def test_stream_ring_needs_an_initialized_upscaler_and_an_open_stream(nsc):
    """nus_upscaler_stream_*: error paths that need no GPU."""
    u = nsc.PyWgpuUpscaler("quality", "nearest")
    with pytest.raises(RuntimeError, match="not initialized"):
        u.stream_open()
    with pytest.raises(RuntimeError, match="no stream is open"):
        u.stream_submit(b"\0" * 16, bytearray(64))
    u.stream_close()  # closing what is not open is not an error


def test_interpolator_validation_without_gpu(nsc):
    it = nsc.WgpuFrameInterpolator("wide")
    assert it.get_last_gpu_duration_ms() is None
    with pytest.raises(ValueError, match=r"Expected 64 bytes per frame for 4x4x4 RGBA, got frame_a: 60 bytes, frame_b: 64 bytes"):
        it.interpolate_py(b"\0" * 60, b"\0" * 64, 4, 4)
    nsc.WgpuFrameInterpolator("no-such-preset")  # defaults to Wide32x8 like the reference


def test_interpolator_trait_shape_without_gpu(nsc):
    """trait FrameInterpolator (interpolation/mod.rs:29-44): name, set_quality / quality, and interpolate before
    initialize -> "Interpolator not initialized" (:368-370)."""
    it = nsc.WgpuFrameInterpolator()
    assert it.name == "HipWarpBlendInterpolator"
    assert it.quality == "medium"
    for q in ("high", "low", "medium"):
        it.set_quality(q)
        assert it.quality == q
    with pytest.raises(RuntimeError, match="unknown interpolation quality"):
        it.set_quality(7)
    with pytest.raises(RuntimeError, match="Interpolator not initialized"):
        it.interpolate(b"\0" * 64, b"\0" * 64, 0.5)
    with pytest.raises(RuntimeError, match="bad dimensions"):
        it.initialize(0, 4)


def test_output_buffers_must_be_writable(nsc):
    """upscale_into / upscale_batch_into / stream_submit refuse read-only outputs: a `bytes` would be mutated in place, any
    other read-only buffer copied and the result lost (raised before any library call: no GPU needed)."""
    u = nsc.PyWgpuUpscaler("quality", "nearest")
    frame = bytes(16)
    for bad in (bytes(64), memoryview(bytearray(64)).toreadonly()):
        with pytest.raises(TypeError, match="writable"):
            u.upscale_into(frame, bad)
        with pytest.raises(TypeError, match="writable"):
            u.upscale_batch_into([frame], [bad])
        with pytest.raises(TypeError, match="writable"):
            u.stream_submit(frame, bad)


def test_fresh_output_bytes_are_built_in_place(nsc):
    """The output `bytes` of upscale / interpolate_py is created with PyBytes_FromStringAndSize(NULL, n) and filled
    before anyone else sees it: a new object per call, of the right size, unrelated to earlier ones."""
    from nu_scaler_amd.upscaler import _out_buffer

    a, ka, addr_a = _out_buffer(1 << 16)
    b, kb, addr_b = _out_buffer(1 << 16)
    assert isinstance(a, bytes) and len(a) == 1 << 16 and a is not b and addr_a != addr_b
    import ctypes

    ctypes.memset(addr_a, 0x5A, len(a))
    assert a == b"\x5a" * (1 << 16)
    assert _out_buffer(0)[0] == b""


@pytest.mark.skipif(os.environ.get("NUS_EXPECT_GPU") == "1", reason="GPU box")
def test_no_cpu_fallback(nsc):
    """Without a HIP device the compute path must fail loudly, never fall back."""
    if nsc.device_count() > 0:
        pytest.skip("a HIP device is present")
    u = nsc.PyWgpuUpscaler("quality", "nearest")
    with pytest.raises(RuntimeError, match="no HIP device"):
        u.initialize(8, 8, 16, 16)
    it = nsc.WgpuFrameInterpolator()
    with pytest.raises(RuntimeError, match="no HIP device"):
        it.interpolate_py(b"\0" * 64, b"\0" * 64, 4, 4)


@pytest.mark.parametrize("n_in,n_out", [(1920, 3840), (1080, 2160), (320, 640), (100, 237), (500, 200), (1000, 210),
                                        (7, 7), (1, 5), (16, 32)])
def test_axis_tables_match_oracle(nsc, oracle_mod, n_in, n_out):
    L = nsc._capi.lib()
    left = np.zeros(n_out, np.int32)
    nt = np.zeros(n_out, np.uint32)
    w = np.zeros((n_out, 32), np.float32)
    r = L.nus_lanczos3_build_axis(n_in, n_out, left.ctypes.data, nt.ctypes.data, w.ctypes.data)
    ol, on, ow = oracle_mod.resize_axis(n_in, n_out)
    assert r == on.max()
    assert np.array_equal(left, ol) and np.array_equal(nt, on) and np.array_equal(w, ow)
    src = np.zeros(n_out, np.uint32)
    assert L.nus_nearest_build_axis(n_in, n_out, src.ctypes.data) == 0
    assert np.array_equal(src, np.minimum(np.arange(n_out, dtype=np.uint64) * n_in // n_out, n_in - 1))
    from oracle import oracle_np as onp

    for variant, clamp in ((0, True), (1, False)):
        i0 = np.zeros(n_out, np.uint32)
        fr = np.zeros(n_out, np.float32)
        assert L.nus_bilinear_build_axis(n_in, n_out, variant, i0.ctypes.data, fr.ctypes.data) == 0
        e0, _, ef = onp._bilinear_coords(n_in, n_out, clamp)
        assert np.array_equal(i0, np.minimum(e0, n_in - 1)) and np.array_equal(fr, ef)


@pytest.mark.parametrize("filt", [0, 1, 2])
@pytest.mark.parametrize("n_in,n_out", [(1920, 3840), (100, 237), (500, 200), (16, 32), (9, 9)])
def test_resize_filter_tables_match_oracle(nsc, oracle_mod, filt, n_in, n_out):
    L = nsc._capi.lib()
    left = np.zeros(n_out, np.int32)
    nt = np.zeros(n_out, np.uint32)
    w = np.zeros((n_out, 32), np.float32)
    r = L.nus_resize_build_axis(filt, n_in, n_out, left.ctypes.data, nt.ctypes.data, w.ctypes.data)
    ol, on, ow = oracle_mod.resize_axis(n_in, n_out, filt)
    assert r == on.max() and np.array_equal(left, ol) and np.array_equal(nt, on) and np.array_equal(w, ow)


def test_lanczos_axis_unsupported_ratio(nsc):
    L = nsc._capi.lib()
    n_out = 10
    left = np.zeros(n_out, np.int32)
    nt = np.zeros(n_out, np.uint32)
    w = np.zeros((n_out, 32), np.float32)
    assert L.nus_lanczos3_build_axis(1000, n_out, left.ctypes.data, nt.ctypes.data, w.ctypes.data) == nsc._capi.ERR_UNSUPPORTED


def test_table_blob_roundtrip_and_validation(nsc):
    blob = nsc.build_tables_blob(320, 240, 640, 480)
    nsc.validate_tables_blob(blob, 320, 240, 640, 480)
    assert blob == nsc.build_tables_blob(320, 240, 640, 480)
    with pytest.raises(ValueError, match="different dimensions"):
        nsc.validate_tables_blob(blob, 320, 240, 641, 480)
    with pytest.raises(ValueError, match="truncated"):
        nsc.validate_tables_blob(blob[:-1], 320, 240, 640, 480)
    bad = bytearray(blob)
    bad[0] ^= 0xFF
    with pytest.raises(ValueError, match="magic"):
        nsc.validate_tables_blob(bytes(bad), 320, 240, 640, 480)
    # corrupt an index so it points outside the source axis
    bad = bytearray(blob)
    off = 8 + 16  # blob header + x-axis header -> first nn_src entry
    bad[off:off + 4] = (10 ** 6).to_bytes(4, "little")
    with pytest.raises(ValueError, match="out of range"):
        nsc.validate_tables_blob(bytes(bad), 320, 240, 640, 480)
    # a tap window that starts left of its predecessor's: x-axis lz_left lives after nn_src, bl_i0, bl_frac
    bad = bytearray(blob)
    off = 8 + 16 + 3 * 640 * 4 + 100 * 4  # lz_left[100]
    bad[off:off + 4] = (0).to_bytes(4, "little")
    with pytest.raises(ValueError, match="backwards"):
        nsc.validate_tables_blob(bytes(bad), 320, 240, 640, 480)
    # every shape the tests use passes the monotonicity check (up, down, ragged, identity)
    for dims in ((50, 31, 127, 64), (48, 27, 20, 11), (1, 1, 5, 3), (7, 5, 7, 5), (3840, 2160, 1920, 1080), (100, 40, 30, 12)):
        for alg in ("lanczos3", "bicubic", "triangle"):
            nsc.validate_tables_blob(nsc.build_tables_blob(*dims, algorithm=alg), *dims)


def test_shard_frames_partition(nsc):
    for n in (0, 1, 7, 300, 301):
        for world in (1, 2, 3, 8):
            chunks = [nsc.shard_frames(n, world, r) for r in range(world)]
            assert sum(c for _, c in chunks) == n
            pos = 0
            for s, c in chunks:
                assert s == pos
                pos += c
            assert max(c for _, c in chunks) - min(c for _, c in chunks) <= 1
    with pytest.raises(ValueError):
        nsc.shard_frames(10, 2, 2)


def test_synthetic_generators_match_oracle(nsc, oracle_mod):
    from nu_scaler_amd import synthetic as syn

    assert np.array_equal(syn.gradient_frame(97, 33, 5), oracle_mod.gen_gradient(97, 33, 5))
    assert np.array_equal(syn.gradient_frame(1920, 1080, 299), oracle_mod.gen_gradient(1920, 1080, 299))
    assert np.array_equal(syn.noise_frame(31, 17, 0x5EED), oracle_mod.gen_noise(31, 17, 0x5EED))
    assert np.array_equal(syn.box_frame(64, 64, (0, 0, 255, 255)), oracle_mod.gen_box(64, 64, (0, 0, 255, 255)))
    import torch

    s = syn.gradient_stream_torch(3, 40, 12, "cpu", first=2).numpy()
    for k in range(3):
        assert np.array_equal(s[k], oracle_mod.gen_gradient(40, 12, 2 + k))


def test_pipeline_unit_accounting(nsc):
    # 107 827 200 B and 26.9568 Mpix per unit at 1080p -> 4K
    fb, ob = 1920 * 1080 * 4, 3840 * 2160 * 4
    assert 3 * fb + 2 * (fb + ob) == 107_827_200
    assert 3 * 1920 * 1080 + 2 * (1920 * 1080 + 3840 * 2160) == 26_956_800


def test_frame_buffer_drop_oldest_and_latest(nsc):
    """Legacy FrameBuffer semantics (Nu_scale/src/capture/frame_buffer.rs:37-55)."""
    fb = nsc.FrameBuffer(capacity=3, max_frame_bytes=4 * 2 * 4)
    assert len(fb) == 0 and fb.capacity == 3 and fb.get_latest_frame() is None and fb.pop_frame() is None
    frames = [bytes([k] * 32) for k in range(5)]
    drops = [fb.add_frame(f, 4, 2) for f in frames]
    assert drops == [0, 0, 0, 1, 2] and len(fb) == 3 and fb.dropped == 2
    data, w, h, seq = fb.get_latest_frame()
    assert (data, w, h, seq) == (frames[4], 4, 2, 4) and len(fb) == 3  # latest stays queued
    assert [fb.pop_frame()[3] for _ in range(3)] == [2, 3, 4] and fb.pop_frame() is None
    with pytest.raises(ValueError):
        fb.add_frame(b"123", 4, 2)


def test_frame_buffer_timeout_and_threads(nsc):
    import threading
    import time

    fb = nsc.FrameBuffer(capacity=5, max_frame_bytes=64)
    t0 = time.perf_counter()
    assert fb.get_latest_frame(timeout_ms=50) is None
    assert time.perf_counter() - t0 >= 0.04
    threading.Timer(0.05, lambda: fb.add_frame(bytes(16), 2, 2)).start()
    got = fb.pop_frame(timeout_ms=2000)
    assert got is not None and got[1:] == (2, 2, 0)


def test_frame_queue_pop_keeps_a_frame_the_buffer_cannot_hold(nsc):
    """pop checks the oldest frame against the caller's buffer under the queue's lock and only then removes it:
    with mixed frame sizes a too-small buffer gets an error and the frame stays queued (the old code
    sized the buffer against the NEWEST frame, then popped the OLDEST)."""
    import ctypes

    from nu_scaler_amd import _capi as C

    L = C.lib()
    q = L.nus_frame_queue_create(4)
    big = bytes(range(256)) * 4      # 16 x 16 RGBA
    small = bytes([7]) * (2 * 2 * 4)  # 2 x 2
    try:
        assert L.nus_frame_queue_add(q, big, 16, 16) == 0
        assert L.nus_frame_queue_add(q, small, 2, 2) == 0
        w, h, seq = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        buf = ctypes.create_string_buffer(64)  # holds the newest (small) frame but not the oldest
        r = L.nus_frame_queue_pop(q, 0, buf, 64, ctypes.byref(w), ctypes.byref(h), ctypes.byref(seq))
        assert r == C.ERR_INVALID_ARGUMENT and b"too small" in L.nus_last_error()
        assert L.nus_frame_queue_size(q) == 2, "the oldest frame must still be queued"
        buf = ctypes.create_string_buffer(1024)
        assert L.nus_frame_queue_pop(q, 0, buf, 1024, ctypes.byref(w), ctypes.byref(h), ctypes.byref(seq)) == 1
        assert (w.value, h.value, seq.value) == (16, 16, 0) and buf.raw[:1024] == big
        assert L.nus_frame_queue_pop(q, 0, buf, 1024, ctypes.byref(w), ctypes.byref(h), ctypes.byref(seq)) == 1
        assert (w.value, h.value, seq.value) == (2, 2, 1) and buf.raw[:16] == small
        assert L.nus_frame_queue_pop(q, 0, buf, 1024, ctypes.byref(w), ctypes.byref(h), ctypes.byref(seq)) == 0
        # a NULL buffer never pops
        assert L.nus_frame_queue_add(q, small, 2, 2) >= 0
        assert L.nus_frame_queue_pop(q, 0, None, 0, None, None, None) == C.ERR_INVALID_ARGUMENT
        assert L.nus_frame_queue_size(q) == 1
    finally:
        L.nus_frame_queue_destroy(q)


def test_no_exception_crosses_the_c_boundary(nsc):
    """An allocation that cannot succeed (a 1 PiB frame copy) comes back as NUS_ERR_OUT_OF_MEMORY, not as a C++
    exception unwinding through a C / Rust / ctypes caller."""
    from nu_scaler_amd import _capi as C

    L = C.lib()
    q = L.nus_frame_queue_create(2)
    try:
        px = bytes(16)
        r = L.nus_frame_queue_add(q, px, 1 << 24, 1 << 24)  # 2^50 bytes: beyond the address space, fails before any copy
        assert r == C.ERR_OUT_OF_MEMORY, r
        assert b"out of memory" in L.nus_last_error()
        assert L.nus_frame_queue_size(q) == 0
    finally:
        L.nus_frame_queue_destroy(q)


def test_header_is_plain_c_and_links(nsc, tmp_path):
    """include/nuscaler_hip.h compiles as strict C99 and a C program linked against libnuscaler_hip.so can drive the
    boundary (tests/c_abi/abi_check.c; no compute calls, so it runs without a GPU)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    lib_dir = os.path.dirname(nsc._capi.LIB_PATH)
    exe = str(tmp_path / "abi_check")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "abi_check.c"), "-o", exe, "-L", lib_dir, "-lnuscaler_hip",
           "-Wl,-rpath," + lib_dir]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0 and "abi_check ok" in run.stdout, run.stdout + run.stderr


def test_copy_pool_stress(tmp_path):
    """The host path's helper threads for the staging copies (csrc/nus_copy.cpp): random sizes and alignments from
    four threads at once, with the pool's default, no and seven workers (tests/c_abi/copy_pool_stress.cpp)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    src_dir = os.path.join(ROOT, "nu_scaler_amd", "csrc")
    exe = str(tmp_path / "copy_pool_stress")
    cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-I", src_dir,
           os.path.join(ROOT, "tests", "c_abi", "copy_pool_stress.cpp"), os.path.join(src_dir, "nus_copy.cpp"), os.path.join(src_dir, "nus_ranges.cpp"), "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    for threads in (None, "0", "7"):
        env = dict(os.environ)
        if threads is not None:
            env["NUS_COPY_THREADS"] = threads
        run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
        assert run.returncode == 0 and "bad 0" in run.stdout, run.stdout + run.stderr
        if threads is not None:
            assert f"workers {threads} " in run.stdout


def _sanitizer_build(tmp_path, name, sources, extra=()):
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    src_dir = os.path.join(ROOT, "nu_scaler_amd", "csrc")
    exe = str(tmp_path / name)
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17", "-pthread", "-I", src_dir,
           *sources, *extra, "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and ("asan" in (res.stderr or "").lower() or "ubsan" in (res.stderr or "").lower()):
        pytest.skip("libasan / libubsan not installed")
    assert res.returncode == 0, res.stderr
    return exe


_SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=67", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1:exitcode=68"}


def test_copy_pool_stress_under_address_and_ub_sanitizers(tmp_path):
    """Round 5: the copy pool and its populate requests under -fsanitize=address,undefined -- buffers shorter
    than a page, buffers given back to the allocator immediately after their wait while other tickets' requests are still queued,
    requests racing with copies into one fresh mapping.  (detect_leaks=0: the pool is never destroyed by design, and the forked
    child of the program has its thread objects but none of its threads.)"""
    import subprocess
    src_dir = os.path.join(ROOT, "nu_scaler_amd", "csrc")
    exe = _sanitizer_build(tmp_path, "copy_pool_stress_asan",
                           [os.path.join(ROOT, "tests", "c_abi", "copy_pool_stress.cpp"), os.path.join(src_dir, "nus_copy.cpp"), os.path.join(src_dir, "nus_ranges.cpp")])
    for threads in ("3", "0"):
        run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, NUS_COPY_THREADS=threads, **_SAN_ENV), timeout=600)
        assert run.returncode == 0 and "bad 0" in run.stdout and "ERROR: AddressSanitizer" not in run.stderr \
            and "runtime error" not in run.stderr, run.stdout + run.stderr


def test_copy_pool_stress_with_two_cpus(tmp_path):
    """The corner an 8-rank job on a 16-CPU box reaches: two CPUs per process leave no CPU for a worker beside the submitting
    and the retiring thread (nus_copy.cpp: n = cpus - 2 = 0) -- every copy is then done by its caller, populate requests are
    declined, and nothing may wait for a worker that does not exist."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or shutil.which("taskset") is None or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("needs g++, taskset and two CPUs")
    src_dir = os.path.join(ROOT, "nu_scaler_amd", "csrc")
    exe = str(tmp_path / "copy_pool_stress")
    res = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", src_dir, os.path.join(ROOT, "tests", "c_abi", "copy_pool_stress.cpp"),
                          os.path.join(src_dir, "nus_copy.cpp"), os.path.join(src_dir, "nus_ranges.cpp"), "-o", exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    cpus = sorted(os.sched_getaffinity(0))[:2]
    env = {k: v for k, v in os.environ.items() if k != "NUS_COPY_THREADS"}
    run = subprocess.run(["taskset", "-c", ",".join(map(str, cpus)), exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0 and "workers 0 bad 0" in run.stdout, run.stdout + run.stderr


def test_host_tables_and_queue_under_address_and_ub_sanitizers(tmp_path):
    """nus_tables.cpp (every builder over a sweep of sizes, ratios and filters, the exact-ratio views, serialisation round trips
    and damaged blobs) and the frame queue, built with -fsanitize=address,undefined (tests/c_abi/host_tables_sanitize.cpp)."""
    import subprocess
    src_dir = os.path.join(ROOT, "nu_scaler_amd", "csrc")
    exe = _sanitizer_build(tmp_path, "host_tables_asan",
                           [os.path.join(ROOT, "tests", "c_abi", "host_tables_sanitize.cpp"), os.path.join(src_dir, "nus_tables.cpp")])
    run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, **_SAN_ENV), timeout=600)
    assert run.returncode == 0 and "bad 0" in run.stdout and "runtime error" not in run.stderr, run.stdout + run.stderr


def test_copy_pool_stress_under_thread_sanitizer(tmp_path):
    """The same program built with -fsanitize=thread (CPU build; GPU sanitizers are not available on this pool): copies, populate
    requests racing with copies into the same fresh mapping, the low-priority queue, worker start-up and shutdown -- no report."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    src_dir = os.path.join(ROOT, "nu_scaler_amd", "csrc")
    exe = str(tmp_path / "copy_pool_stress_tsan")
    cmd = ["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-pthread", "-I", src_dir,
           os.path.join(ROOT, "tests", "c_abi", "copy_pool_stress.cpp"), os.path.join(src_dir, "nus_copy.cpp"), os.path.join(src_dir, "nus_ranges.cpp"), "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and "tsan" in (res.stderr or "").lower():
        pytest.skip("libtsan not installed")
    assert res.returncode == 0, res.stderr
    env = dict(os.environ, NUS_COPY_THREADS="3", TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    if "FATAL: ThreadSanitizer" in run.stderr and "memory layout" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert run.returncode == 0 and "bad 0" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr, run.stdout + run.stderr


def test_vram_stats_surface(nsc):
    """PyAdvancedWgpuUpscaler.get_vram_stats / get_vram_usage_percent (lib.rs:539-584): present; without a GPU
    they raise the reference's RuntimeError text, with one they report hipMemGetInfo."""
    u = nsc.create_advanced_upscaler("quality")
    assert hasattr(u, "get_vram_stats") and hasattr(u, "get_vram_usage_percent")
    if nsc.device_count() == 0:
        with pytest.raises(RuntimeError, match="No GPU resources available"):
            u.get_vram_stats()
    else:
        st = u.get_vram_stats()
        assert st.total_mb > 100_000 and 0.0 <= st.usage_percent <= 100.0
        assert abs(st.total_mb - st.used_mb - st.free_mb) < 1.0
        assert u.get_vram_usage_percent() == pytest.approx(st.usage_percent, abs=5.0)
    s = nsc.PyVramStats(100.0, 25.0, 75.0, 1.0)
    assert s.usage_percent == 25.0


def test_benchmark_api_surface(nsc):
    """py_benchmark_upscaler / py_run_comparison_benchmark / PyBenchmarkResult (benchmark.rs:24-272): argument
    defaults, output size rounding and error text; without a GPU the run fails loudly with the reference's prefix."""
    from nu_scaler_amd import benchmark as b
    assert b._round_half_away(2.5) == 3 and b._round_half_away(3.5) == 4 and b._round_half_away(479.4) == 479
    if nsc.device_count() == 0:
        with pytest.raises(RuntimeError, match="^Benchmark error: "):
            nsc.py_benchmark_upscaler("wgpu", "quality", 64, 64, 2.0, 2)
        assert nsc.py_run_comparison_benchmark(32, 32, 2.0, 1) == []
    r = b.PyBenchmarkResult(upscaler_name="x", technology="Wgpu", quality="Quality", input_width=1, input_height=2,
                            output_width=3, output_height=4, scale_factor=2.0, avg_frame_time_ms=1.0, fps=1000.0,
                            frames_processed=5, total_duration_ms=5.0)
    assert (r.input_width, r.output_height, r.frames_processed, r.technology) == (1, 4, 5, "Wgpu")


def test_transfer_module_without_a_gpu(nsc):
    """nu_scaler_amd/transfer.py (nus_download / nus_upload): CPU tensors pass through untouched, and with no HIP device the
    entry points refuse loudly -- there is no fallback to a runtime copy or to torch."""
    import torch

    from nu_scaler_amd import _capi, transfer

    t = torch.arange(1 << 18, dtype=torch.uint8)
    a = transfer.to_numpy(t)
    assert a.base is not None or a.ctypes.data == t.data_ptr()  # the tensor's own memory, no copy
    assert int(a.sum()) == int(t.sum())
    L = _capi.lib()
    assert L.nus_download(None, None, 0, None) == _capi.OK  # nothing to move
    assert L.nus_download(None, None, 16, None) == _capi.ERR_INVALID_ARGUMENT and "null pointer" in _capi.last_error()
    assert L.nus_upload(None, None, 16, None) == _capi.ERR_INVALID_ARGUMENT
    if nsc.device_count() == 0:
        buf = bytearray(64)
        with pytest.raises(RuntimeError, match="no HIP device"):
            transfer.download(0x1000, 64, buf)
        with pytest.raises(RuntimeError, match="no HIP device"):
            transfer.upload(0x1000, buf)
    with pytest.raises(ValueError):
        transfer.download(0x1000, 65, bytearray(64))
    with pytest.raises(TypeError):
        transfer.download(0x1000, 4, b"abcd")  # read-only output


def test_unpin_of_a_pointer_that_was_never_pinned_is_refused(nsc):
    """nus_host_unpin asks the library's own record (nus_host_ranges) before the runtime: an arbitrary pointer is an
    invalid argument, not a call into hipHostUnregister."""
    import ctypes

    from nu_scaler_amd import _capi

    b = bytearray(4096)
    a = (ctypes.c_ubyte * 4096).from_buffer(b)
    L = _capi.lib()
    assert L.nus_host_unpin(ctypes.addressof(a)) == _capi.ERR_INVALID_ARGUMENT
    assert "not the start of a buffer pinned with nus_host_pin" in _capi.last_error()
    assert L.nus_host_unpin(None) == _capi.ERR_INVALID_ARGUMENT


def test_no_replacement_of_tensor_cpu_anywhere():
    """the harness fetches device tensors through the product's nus_download (`fetch`), the product package
    holds no assignment to torch.Tensor.cpu, and no test / bench / smoke code calls `.cpu()` on what may be a device tensor."""
    import glob
    import re

    pkg = glob.glob(os.path.join(ROOT, "nu_scaler_amd", "*.py"))
    for p in pkg + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py"), os.path.join(ROOT, "tests", "conftest.py")]:
        src = open(p).read()
        assert not re.search(r"Tensor\.cpu\s*=", src), p
        assert "route_tensor_cpu" not in src, p
    allowed = {("stream.py", "g.cpu().tolist()")}  # one gathered row of float64 numbers (a few hundred bytes)
    for p in pkg + glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "helpers", "*.py")) + \
            [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        if os.path.basename(p) == "test_host_logic.py":
            continue
        for ln in open(p).read().splitlines():
            code = ln.split("#", 1)[0]
            if ".cpu()" in code:
                assert any(os.path.basename(p) == f and frag in code for f, frag in allowed), (p, ln)


def test_fatal_trace_report_of_a_dying_child(nsc, tmp_path):
    """nus_install_fatal_trace: a child that aborts leaves, in this order, the native frames of the raising thread, the library's
    record of host ranges and /proc/self/maps (with its heap line) on the descriptor it named, then dies of SIGABRT as before."""
    import signal
    import subprocess

    code = ("import os, sys\nsys.path.insert(0, %r)\nimport nu_scaler_amd as n\n"
            "assert n.install_fatal_trace(2) and n.install_fatal_trace(2)\nos.abort()\n" % ROOT)
    import sys

    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert run.returncode == -signal.SIGABRT, (run.returncode, run.stderr[-2000:])
    err = run.stderr
    marks = ["[nus_fatal_trace] fatal signal SIGABRT", "[nus_fatal_trace] end of native frames",
             "[nus_fatal_trace] host ranges the library holds now: 0", "[nus_fatal_trace] last 0 range events",
             "[nus_fatal_trace] /proc/self/maps", "[heap]", "[nus_fatal_trace] end of /proc/self/maps"]
    pos = [err.find(m) for m in marks]
    assert all(p >= 0 for p in pos) and pos == sorted(pos), pos
    assert "abort" in err[pos[0]:pos[1]] and "libnuscaler_hip.so" in err[pos[4]:pos[6]]


def test_guard_registry_of_the_tests_is_the_one_the_teardown_checks(request):
    """tests/conftest.py `guarded`: the object the test modules import is the object the autouse fixture checks after every test (a
    second copy of conftest -- another import mode -- would leave every guard band unchecked without anybody noticing)."""
    from conftest import guarded

    plugins = [m for m in request.config.pluginmanager.get_plugins() if getattr(m, "__name__", "") == "conftest"]
    assert plugins and all(getattr(m, "guarded") is guarded for m in plugins)
    # and a damaged guard is reported (host tensors: the check itself needs no GPU)
    import torch

    t = guarded.zeros((4, 8), dtype=torch.int16, device="cpu")
    assert t.shape == (4, 8) and int(t.abs().sum()) == 0
    flat, n = guarded._live[-1]
    assert n == 64 and flat.numel() == 2 * guarded.GUARD + 64 + 192
    flat[guarded.GUARD - 1] = 0
    torch_cuda_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda: None  # (no device here)
    try:
        with pytest.raises(AssertionError, match="1 bytes in front"):
            guarded.assert_intact()
    finally:
        torch.cuda.synchronize = torch_cuda_sync
