"""ctypes binding of include/nuscaler_hip.h -- the same entry points a Rust shim
(`nu_scaler_hip-sys`) would bind.  No torch types cross here."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NUS_LIB_PATH: dev override (timing-only A/B builds, tools/lz_ab.sh)
LIB_PATH = os.environ.get("NUS_LIB_PATH") or os.path.join(_HERE, "lib", "libnuscaler_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

# nus_status
OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_NOT_INITIALIZED = -2
ERR_SIZE_MISMATCH = -3
ERR_HIP = -4
ERR_NO_DEVICE = -5
ERR_UNSUPPORTED = -6
ERR_OUT_OF_MEMORY = -7

ALG_NEAREST, ALG_BILINEAR, ALG_LANCZOS3, ALG_BICUBIC, ALG_TRIANGLE = 0, 1, 2, 3, 4
ALG_FSR1, ALG_FSR_EASU, ALG_FSR_RCAS = 5, 6, 7
FORMAT_RGBA8, FORMAT_BGRA8, FORMAT_RGBX8, FORMAT_BGRX8 = 0, 1, 2, 3
QUALITY_ULTRA_PERFORMANCE, QUALITY_ULTRA, QUALITY_QUALITY, QUALITY_BALANCED, QUALITY_PERFORMANCE, QUALITY_NATIVE = range(6)
TECH_NONE, TECH_FSR, TECH_DLSS, TECH_WGPU, TECH_FALLBACK = range(5)
WG_SQUARE_8X8, WG_SQUARE_16X16, WG_WIDE_32X8, WG_TALL_8X32 = range(4)
RESIZE_MAX_TAPS = 32

# Every symbol include/nuscaler_hip.h declares: (name, restype, argtypes)
_vp, _cp, _i, _u32, _sz, _i64 = (ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32,
                                 ctypes.c_size_t, ctypes.c_int64)
_f, _dp = ctypes.c_float, ctypes.POINTER(ctypes.c_double)
SIGNATURES = [
    ("nus_abi_version", _i, []),
    ("nus_device_count", _i, []),
    ("nus_device_memory_info", _i, [_i, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("nus_host_pin", _i, [_vp, _sz]),
    ("nus_host_unpin", _i, [_vp]),
    ("nus_download", _i, [_vp, _vp, _sz, _vp]),
    ("nus_upload", _i, [_vp, _vp, _sz, _vp]),
    ("nus_host_ranges", _sz, [_vp, _sz, _i]),
    ("nus_install_fatal_trace", _i, [_i]),
    ("nus_last_error", _cp, []),
    ("nus_status_string", _cp, [_i]),
    ("nus_upscaler_create", _vp, [_i, _i]),
    ("nus_upscaler_create_for_technology", _vp, [_i, _i]),
    ("nus_upscaler_destroy", None, [_vp]),
    ("nus_upscaler_set_device", _i, [_vp, _i]),
    ("nus_upscaler_set_bilinear_variant", _i, [_vp, _i]),
    ("nus_upscaler_set_lanczos_mode", _i, [_vp, _i]),
    ("nus_upscaler_set_option", _i, [_vp, _cp, _i64]),
    ("nus_upscaler_get_option", _i, [_vp, _cp, ctypes.POINTER(_i64)]),
    ("nus_upscaler_set_input_format", _i, [_vp, _i]),
    ("nus_upscaler_set_sharpness", _i, [_vp, _f, _f]),
    ("nus_upscaler_get_sharpness", _i, [_vp, ctypes.POINTER(_f), ctypes.POINTER(_f)]),
    ("nus_upscaler_initialize", _i, [_vp, _u32, _u32, _u32, _u32]),
    ("nus_upscaler_upscale", _i, [_vp, _vp, _sz, _vp, _sz]),
    ("nus_upscaler_upscale_batch", _i, [_vp, _vp, _vp, _sz, _vp, _sz]),
    ("nus_upscaler_stream_open", _i, [_vp]),
    ("nus_upscaler_stream_submit", _i, [_vp, _vp, _sz, _vp, _sz, ctypes.POINTER(ctypes.c_uint64)]),
    ("nus_upscaler_stream_wait", _i, [_vp, ctypes.c_uint64]),
    ("nus_upscaler_stream_close", _i, [_vp]),
    ("nus_upscaler_upscale_device", _i, [_vp, _vp, _vp, _u32, _vp]),
    ("nus_upscaler_upscale_blend_device", _i, [_vp, _vp, _sz, _vp, _sz, _f, _vp, _u32, _vp]),
    ("nus_upscaler_upscale_unit_device", _i, [_vp, _vp, _sz, _vp, _sz, _f, _vp, _vp, _vp, _u32, _vp]),
    ("nus_upscaler_name", _cp, [_vp]),
    ("nus_upscaler_algorithm", _i, [_vp]),
    ("nus_upscaler_quality", _i, [_vp]),
    ("nus_upscaler_set_quality", _i, [_vp, _i]),
    ("nus_upscaler_is_initialized", _i, [_vp]),
    ("nus_upscaler_input_size", _sz, [_vp]),
    ("nus_upscaler_output_size", _sz, [_vp]),
    ("nus_upscaler_last_error", _cp, [_vp]),
    ("nus_upscaler_last_gpu_ms", _i, [_vp, _dp]),
    ("nus_upscaler_set_profiling", _i, [_vp, _i]),
    ("nus_upscaler_profile_collect", _i, [_vp, ctypes.POINTER(ctypes.c_uint64), _dp]),
    ("nus_upscaler_kernel_variant", _cp, [_vp]),
    ("nus_upscaler_export_tables", _i64, [_vp, _vp, _sz]),
    ("nus_upscaler_import_tables", _i, [_vp, _vp, _sz]),
    ("nus_tables_build_blob", _i64, [_u32, _u32, _u32, _u32, _i, _vp, _sz]),
    ("nus_tables_build_blob_for", _i64, [_i, _u32, _u32, _u32, _u32, _i, _vp, _sz]),
    ("nus_tables_validate_blob", _i, [_vp, _sz, _u32, _u32, _u32, _u32]),
    ("nus_lanczos3_build_axis", _i, [_u32, _u32, _vp, _vp, _vp]),
    ("nus_resize_build_axis", _i, [_i, _u32, _u32, _vp, _vp, _vp]),
    ("nus_nearest_build_axis", _i, [_u32, _u32, _vp]),
    ("nus_bilinear_build_axis", _i, [_u32, _u32, _i, _vp, _vp]),
    ("nus_interp_create", _vp, [_i]),
    ("nus_interp_destroy", None, [_vp]),
    ("nus_interp_set_device", _i, [_vp, _i]),
    ("nus_interp_set_input_format", _i, [_vp, _i]),
    ("nus_interp_set_flow_format", _i, [_vp, _i]),
    ("nus_interp_set_mode", _i, [_vp, _i]),
    ("nus_interp_mode", _i, [_vp]),
    ("nus_interp_initialize", _i, [_vp, _u32, _u32]),
    ("nus_interp_interpolate_frames", _i, [_vp, _vp, _sz, _vp, _sz, _f, _vp, _sz]),
    ("nus_interp_name", _cp, [_vp]),
    ("nus_interp_set_quality", _i, [_vp, _i]),
    ("nus_interp_quality", _i, [_vp]),
    ("nus_interp_interpolate", _i, [_vp, _vp, _sz, _vp, _sz, _vp, _u32, _u32, _f, _vp, _sz]),
    ("nus_interp_interpolate_device", _i, [_vp, _vp, _sz, _vp, _sz, _vp, _u32, _u32, _f, _vp, _u32, _vp]),
    ("nus_interp_last_gpu_ms", _i, [_vp, _dp]),
    ("nus_interp_last_error", _cp, [_vp]),
    ("nus_frame_queue_create", _vp, [_sz]),
    ("nus_frame_queue_destroy", None, [_vp]),
    ("nus_frame_queue_add", _i64, [_vp, _vp, _u32, _u32]),
    ("nus_frame_queue_latest", _i, [_vp, _i64, _vp, _sz, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(ctypes.c_uint64)]),
    ("nus_frame_queue_pop", _i, [_vp, _i64, _vp, _sz, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(ctypes.c_uint64)]),
    ("nus_frame_queue_size", _sz, [_vp]),
    ("nus_frame_queue_capacity", _sz, [_vp]),
    ("nus_frame_queue_dropped", ctypes.c_uint64, [_vp]),
    ("nus_probe_device", _i, [_i, _vp, _vp, _sz, _u32, _vp]),
    ("nus_host_pending_pieces", _sz, []),
    ("nus_swizzle_bgra_to_rgba_device", _i, [_vp, _vp, _sz, _vp]),
    ("nus_flow_create", _vp, []),
    ("nus_flow_destroy", None, [_vp]),
    ("nus_flow_set_device", _i, [_vp, _i]),
    ("nus_flow_set_tiled", _i, [_vp, _i]),
    ("nus_flow_set_mode", _i, [_vp, _i]),
    ("nus_flow_mode", _i, [_vp]),
    ("nus_flow_last_error", _cp, [_vp]),
    ("nus_flow_rgba8_to_f32", _i, [_vp, _vp, _u32, _u32, _vp]),
    ("nus_flow_blur", _i, [_vp, _vp, _u32, _u32, _vp]),
    ("nus_flow_downsample", _i, [_vp, _vp, _u32, _u32, _vp]),
    ("nus_flow_horn_schunck", _i, [_vp, _vp, _vp, _vp, _u32, _u32, _f, _u32, _vp]),
    ("nus_flow_upsample", _i, [_vp, _vp, _u32, _u32, _vp, _u32, _u32, _f]),
    ("nus_flow_estimate", _i, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f, _vp]),
    ("nus_flow_estimate_device", _i, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f, _vp, _vp]),
    ("nus_flow_estimate_device_stream", _i, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _f, _vp, _vp]),
    ("nus_flow_interpolate_device_stream", _i, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _f, _f, _i, _vp, _vp, _vp]),
]


class NuScalerLibraryError(ImportError):
    """libnuscaler_hip.so is missing or incomplete.  There is no CPU fallback."""


def _source_digest() -> str:
    """sha256 over the contents of every file the library is built from, in the order of the Makefile's
    SRCS_ALL (content, not mtimes: a fresh checkout gives every file the same mtime, so timestamps cannot
    tell a stale .so from a current one).  The Makefile writes the same digest next to the .so."""
    import glob
    import hashlib

    names = []
    for pat in ("*.hip", "*.cpp", "*.hpp", "cli/*.cpp", "cli/*.hpp"):
        names += [os.path.relpath(f, CSRC_DIR) for f in glob.glob(os.path.join(CSRC_DIR, pat))]
    names += ["Makefile", "../../include/nuscaler_hip.h"]
    h = hashlib.sha256()
    for n in sorted(set(names)):
        with open(os.path.join(CSRC_DIR, n), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """Compile libnuscaler_hip.so for gfx950 with hipcc (works without a GPU).  The library is
    rebuilt from scratch (`make -B`) unless the digest of the sources it was built from, kept
    next to it, matches the sources on disk."""
    stamp = LIB_PATH + ".srcsha256"
    digest = _source_digest()
    built_from = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or not os.path.exists(LIB_PATH) or built_from != digest:
        res = subprocess.run(["make", "-j8", "-B", "-C", CSRC_DIR], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("building libnuscaler_hip.so failed:\n" + res.stdout + res.stderr)
        if open(stamp).read().strip() != digest:
            raise RuntimeError("Makefile and nu_scaler_amd._capi disagree on the source list (SRCS_ALL)")
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    """Load the C-ABI library; fail loudly if it is absent (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NuScalerLibraryError(
                f"{LIB_PATH} not found: build it with `make -C {CSRC_DIR}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "nu_scaler_amd has no CPU fallback.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise NuScalerLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, res, args in SIGNATURES:
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise NuScalerLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if L.nus_abi_version() != 1:
            raise NuScalerLibraryError("libnuscaler_hip.so ABI version mismatch")
        _lib = L
    return _lib


def last_error() -> str:
    return lib().nus_last_error().decode("utf-8", "replace")


def device_count() -> int:
    return int(lib().nus_device_count())


class PinnedBuffer:
    """A writable buffer (bytearray, numpy array, ...) pinned for the DMA engines while the object lives / inside a `with`:
    the host entry points then copy straight from / into it (nus_host_pin / nus_host_unpin).  Keep the object as long as the
    buffer is handed to upscale / upscale_batch / interpolate calls; do not resize the buffer meanwhile."""

    _addr = None  # (so that __del__ of an object whose __init__ raised early finds the attribute)

    def __init__(self, buffer):
        mv = memoryview(buffer)
        if mv.readonly or not mv.c_contiguous:
            raise TypeError("PinnedBuffer needs a writable C-contiguous buffer")
        self.buffer = buffer
        self._arr = (ctypes.c_ubyte * mv.nbytes).from_buffer(mv.cast("B"))
        self._addr = ctypes.addressof(self._arr)
        if lib().nus_host_pin(self._addr, mv.nbytes) != OK:
            self._addr = None
            raise RuntimeError(last_error())

    def unpin(self) -> None:
        """Raises if the runtime refuses: a registration that outlives its buffer must not go unnoticed (the buffer is kept
        alive by this object in that case)."""
        addr = self._addr
        if addr is not None:
            if lib().nus_host_unpin(addr) != OK:
                raise RuntimeError(last_error())
            self._addr = None
            self._arr = None

    def __enter__(self):
        return self.buffer

    def __exit__(self, *exc):
        self.unpin()

    def __del__(self):
        self.unpin()  # (a failure here is printed by the interpreter as "Exception ignored in ...": loud on purpose)
