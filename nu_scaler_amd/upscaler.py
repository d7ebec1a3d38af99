"""Python mirror of the reference's pyo3 upscaler classes, bound to the HIP C ABI.

Same class names, constructor strings, methods and error behaviour as
nu_scaler_core/src/lib.rs:39-160 (PyWgpuUpscaler) and :328-729
(PyAdvancedWgpuUpscaler), so scripts written against `nu_scaler_core` run with
`import nu_scaler_amd as nu_scaler_core`.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List, Optional

from . import _capi as C

_QUALITY = {  # lib.rs:51-57; unknown strings silently become "quality"
    "ultra": C.QUALITY_ULTRA,
    "quality": C.QUALITY_QUALITY,
    "balanced": C.QUALITY_BALANCED,
    "performance": C.QUALITY_PERFORMANCE,
}
_QUALITY_STR = {v: k for k, v in _QUALITY.items()}
_ALGORITHM = {  # lib.rs:58-62; unknown strings silently become "nearest"
    "nearest": C.ALG_NEAREST,
    "bilinear": C.ALG_BILINEAR,
    # new value of this build (`"lanczos3"` added)
    "lanczos3": C.ALG_LANCZOS3,
    "lanczos": C.ALG_LANCZOS3,
    # image-0.24.9 filters of the legacy BasicUpscaler (Nu_scale/src/upscale/common.rs:233-260)
    "bicubic": C.ALG_BICUBIC,
    "catmullrom": C.ALG_BICUBIC,
    "triangle": C.ALG_TRIANGLE,
    # FSR1-style shader pair of nu_scaler_core/src/upscale/fsr.rs:24-260
    "fsr1": C.ALG_FSR1,
    "fsr": C.ALG_FSR1,
    "easu": C.ALG_FSR_EASU,
    "rcas": C.ALG_FSR_RCAS,
}


def _as_buffer(data):
    """Borrow a readable bytes-like object without copying: (address, length, keepalive)."""
    mv = memoryview(data)
    if not mv.c_contiguous:
        raise TypeError("frame buffer must be C-contiguous")
    mv = mv.cast("B")
    if mv.readonly:
        # bytes: from_buffer needs a writable view; c_char_p borrows the storage instead
        keep = bytes(data) if not isinstance(data, bytes) else data
        return ctypes.cast(ctypes.c_char_p(keep), ctypes.c_void_p).value, len(keep), keep
    arr = (ctypes.c_ubyte * len(mv)).from_buffer(mv)
    return ctypes.addressof(arr), len(mv), (arr, mv)


def _as_out_buffer(out, need: int):
    """Borrow a caller-provided OUTPUT buffer: it must be writable (a `bytes` would be mutated in place, any other read-only
    buffer would be copied and the result lost in the temporary) and hold at least `need` bytes."""
    mv = memoryview(out)
    if mv.readonly:
        raise TypeError("output buffer must be writable (bytearray, numpy array, PinnedBuffer, ...), not " + type(out).__name__)
    if not mv.c_contiguous:
        raise TypeError("output buffer must be C-contiguous")
    mv = mv.cast("B")
    if len(mv) < need:
        raise ValueError(f"output buffer holds {len(mv)} bytes, the output frame needs {need}")
    arr = (ctypes.c_ubyte * len(mv)).from_buffer(mv)
    return ctypes.addressof(arr), len(mv), (arr, mv)


_PyBytes_FromStringAndSize = ctypes.pythonapi.PyBytes_FromStringAndSize
_PyBytes_FromStringAndSize.restype = ctypes.py_object
_PyBytes_FromStringAndSize.argtypes = [ctypes.c_char_p, ctypes.c_ssize_t]
_PyBytes_AsString = ctypes.pythonapi.PyBytes_AsString
_PyBytes_AsString.restype = ctypes.c_void_p
_PyBytes_AsString.argtypes = [ctypes.py_object]


def _out_buffer(size: int):
    """Fresh output `bytes` the C side fills: (bytes, keepalive, address).  The CPython idiom for building a bytes object in
    place -- PyBytes_FromStringAndSize(NULL, n) returns an uninitialised object "whose contents may be filled in" through
    PyBytes_AsString as long as nobody else holds a reference yet -- which is what pyo3's PyBytes::new_with does for the
    reference's own return value; no second 33 MB copy, and no write through a pointer to an object that is already shared."""
    if size == 0:
        return b"", None, None
    out = _PyBytes_FromStringAndSize(None, size)
    return out, out, _PyBytes_AsString(out)


class PyWgpuUpscaler:
    """lib.rs:39-160.  `PyWgpuUpscaler(quality="quality", algorithm="nearest")`."""

    def __init__(self, quality: str = "quality", algorithm: str = "nearest", *, device: int = 0,
                 bilinear_variant: str = "cpu", lanczos_mode: str = "fma"):
        self._lib = C.lib()
        q = _QUALITY.get(str(quality).lower(), C.QUALITY_QUALITY)
        a = _ALGORITHM.get(str(algorithm).lower(), C.ALG_NEAREST)
        self._h = self._lib.nus_upscaler_create(a, q)
        if not self._h:
            raise RuntimeError(C.last_error())
        self._upscale_scale = 2.0  # lib.rs:65
        self._device = int(device)
        self._check(self._lib.nus_upscaler_set_device(self._h, int(device)))
        self._check(self._lib.nus_upscaler_set_bilinear_variant(self._h, 1 if bilinear_variant == "wgsl" else 0))
        self._check(self._lib.nus_upscaler_set_lanczos_mode(self._h, 1 if lanczos_mode == "exact" else 0))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.nus_upscaler_destroy(h)

    # -- error mapping: anyhow -> PyRuntimeError(e.to_string()) (lib.rs:85, :110)
    def _check(self, status: int) -> None:
        if status != C.OK:
            raise RuntimeError(self._lib.nus_upscaler_last_error(self._h).decode("utf-8", "replace"))

    def initialize(self, input_width: int, input_height: int, output_width: int, output_height: int) -> None:
        if input_width > 0 and input_height > 0:  # lib.rs:77-81
            self._upscale_scale = (output_width / input_width + output_height / input_height) / 2.0
        self._check(self._lib.nus_upscaler_initialize(self._h, input_width, input_height, output_width, output_height))

    @property
    def upscale_scale(self) -> float:
        return self._upscale_scale

    @upscale_scale.setter
    def upscale_scale(self, scale: float) -> None:
        if scale < 1.0 or scale > 4.0:  # lib.rs:95-99
            raise ValueError("Scale factor must be between 1.0 and 4.0")
        self._upscale_scale = float(scale)

    def upscale(self, input) -> bytes:
        """bytes in, bytes out (lib.rs:105-112)."""
        addr, n, keep = _as_buffer(input)
        out_size = self._lib.nus_upscaler_output_size(self._h)
        out, oarr, oaddr = _out_buffer(out_size)
        self._check(self._lib.nus_upscaler_upscale(self._h, addr, n, oaddr, out_size))
        del oarr, keep
        return out

    def upscale_into(self, input, out) -> None:
        """Zero-copy variant: writes into a caller-provided writable buffer."""
        addr, n, keep = _as_buffer(input)
        oaddr, on, okeep = _as_out_buffer(out, self.output_size)
        self._check(self._lib.nus_upscaler_upscale(self._h, addr, n, oaddr, on))
        del keep, okeep

    def upscale_batch(self, frames: Iterable) -> List[bytes]:
        """lib.rs:140-154; frames are pipelined over copy/compute streams on the GPU."""
        bufs = [_as_buffer(f) for f in frames]
        n = len(bufs)
        out_size = self._lib.nus_upscaler_output_size(self._h)
        if n == 0:  # still reports "not initialized" like upscale/mod.rs:610-614
            self._check(self._lib.nus_upscaler_upscale_batch(self._h, None, None, 0, None, 0))
            return []
        triples = [_out_buffer(out_size) for _ in range(n)]
        ins_c = (ctypes.c_void_p * n)(*[b[0] for b in bufs])
        lens_c = (ctypes.c_size_t * n)(*[b[1] for b in bufs])
        outs_c = (ctypes.c_void_p * n)(*[t[2] for t in triples])
        self._check(self._lib.nus_upscaler_upscale_batch(self._h, ins_c, lens_c, n, outs_c, out_size))
        outs = [t[0] for t in triples]
        del triples
        return outs

    def upscale_batch_into(self, frames: Iterable, outs: Iterable) -> None:
        """`upscale_batch` into caller-provided writable buffers (one per frame, each >= output_size): what a Rust caller
        of nus_upscaler_upscale_batch does with its own Vecs; no allocation on the way."""
        bufs = [_as_buffer(f) for f in frames]
        need = self.output_size
        obufs = [_as_out_buffer(o, need) for o in outs]
        n = len(bufs)
        if len(obufs) != n:
            raise ValueError("upscale_batch_into: one output buffer per frame")
        if n == 0:
            return
        ins_c = (ctypes.c_void_p * n)(*[b[0] for b in bufs])
        lens_c = (ctypes.c_size_t * n)(*[b[1] for b in bufs])
        outs_c = (ctypes.c_void_p * n)(*[b[0] for b in obufs])
        self._check(self._lib.nus_upscaler_upscale_batch(self._h, ins_c, lens_c, n, outs_c, min(b[1] for b in obufs)))
        del bufs, obufs

    # -- persistent ring (nus_upscaler_stream_*): frames in one at a time, results in order, three in flight
    def stream_open(self) -> None:
        self._stream_keep = {}
        self._check(self._lib.nus_upscaler_stream_open(self._h))

    def stream_submit(self, frame, out) -> int:
        """Stage and enqueue one frame; `out` is a writable buffer of output_size bytes that receives it.  Returns the frame's
        ticket; blocks only while three frames are in flight."""
        addr, n, keep = _as_buffer(frame)
        oaddr, on, okeep = _as_out_buffer(out, self.output_size)
        t = ctypes.c_uint64()
        self._check(self._lib.nus_upscaler_stream_submit(self._h, addr, n, oaddr, on, ctypes.byref(t)))
        self._stream_keep[t.value] = (keep, okeep)  # both buffers stay alive until the frame has been waited for
        return t.value

    def stream_wait(self, ticket: int) -> None:
        st = self._lib.nus_upscaler_stream_wait(self._h, int(ticket))
        for k in [k for k in self._stream_keep if k <= ticket]:
            del self._stream_keep[k]
        if st != C.OK:
            raise RuntimeError(C.last_error())

    def stream_close(self) -> None:
        st = self._lib.nus_upscaler_stream_close(self._h)
        self._stream_keep = {}
        self._check(st)

    # -- device-resident path (not in the reference; used by the frame stream + bench)
    def upscale_device(self, d_in: int, d_out: int, n_frames: int = 1, stream: int = 0) -> None:
        self._check(self._lib.nus_upscaler_upscale_device(self._h, d_in, d_out, n_frames, stream or None))

    def upscale_blend_device(self, d_a: int, a_stride: int, d_b: int, b_stride: int, time_t: float, d_out: int,
                             n_frames: int = 1, stream: int = 0) -> None:
        """Fused zero-flow interpolate + upscale of the in-between frame (exact-x2 resize kernels)."""
        self._check(self._lib.nus_upscaler_upscale_blend_device(self._h, d_a, a_stride, d_b, b_stride, float(time_t),
                                                                d_out, n_frames, stream or None))

    def upscale_unit_device(self, d_a: int, a_stride: int, d_b: int, b_stride: int, time_t: float, d_mid: int,
                            d_out_real: int, d_out_mid: int, n_units: int = 1, stream: int = 0) -> None:
        """One pipeline step in one launch: upscale(A), blend(A, B) (d_mid, 0 = not wanted) and upscale(blend(A, B))."""
        self._check(self._lib.nus_upscaler_upscale_unit_device(self._h, d_a, a_stride, d_b, b_stride, float(time_t),
                                                               d_mid or None, d_out_real, d_out_mid, n_units, stream or None))

    # -- wgpu-only knobs: accepted and ignored (lib.rs:115-137)
    def reload_shader(self, path: str) -> None:
        return None

    def set_thread_count(self, n: int) -> None:
        return None

    def set_buffer_pool_size(self, n: int) -> None:
        return None

    def set_gpu_allocator(self, preset: str) -> None:
        return None

    @property
    def name(self) -> str:
        return self._lib.nus_upscaler_name(self._h).decode()

    # -- extras
    @property
    def kernel_variant(self) -> str:
        return self._lib.nus_upscaler_kernel_variant(self._h).decode()

    def set_option(self, key: str, value: int) -> None:
        self._check(self._lib.nus_upscaler_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        """What the library decided ("pq_p", "pq_q", "pq_narrow_active", "rows_per_wave": nus_upscaler_get_option)."""
        import ctypes

        v = ctypes.c_int64(0)
        self._check(self._lib.nus_upscaler_get_option(self._h, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def set_input_format(self, fmt: str) -> None:
        """"rgba" (default) or "bgra": captured frames are swizzled inside the kernels' loads
        (the reference's CPU loop, lib.rs:251-270).  The output is always RGBA."""
        f = {"rgba": C.FORMAT_RGBA8, "bgra": C.FORMAT_BGRA8, "rgbx": C.FORMAT_RGBX8, "bgrx": C.FORMAT_BGRX8}.get(str(fmt).lower())
        if f is None:
            raise ValueError("input format must be 'rgba', 'bgra', 'rgbx' or 'bgrx'")
        self._check(self._lib.nus_upscaler_set_input_format(self._h, f))

    def set_sharpness(self, easu: float = -1.0, rcas: float = -1.0) -> None:
        """FSR1-style passes: shader `sharpness` uniforms; negative keeps the quality default."""
        self._check(self._lib.nus_upscaler_set_sharpness(self._h, float(easu), float(rcas)))

    def get_sharpness(self):
        e, r = ctypes.c_float(), ctypes.c_float()
        self._check(self._lib.nus_upscaler_get_sharpness(self._h, ctypes.byref(e), ctypes.byref(r)))
        return e.value, r.value

    def set_lanczos_mode(self, mode: str) -> None:
        self._check(self._lib.nus_upscaler_set_lanczos_mode(self._h, 1 if mode == "exact" else 0))

    def set_profiling(self, enabled: bool) -> None:
        self._check(self._lib.nus_upscaler_set_profiling(self._h, int(bool(enabled))))

    def profile_collect(self):
        """(launches, total_ms) of the main-kernel hipEvent pairs recorded since the last call."""
        n, ms = ctypes.c_uint64(), ctypes.c_double()
        self._check(self._lib.nus_upscaler_profile_collect(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return int(n.value), float(ms.value)

    def get_last_gpu_duration_ms(self) -> Optional[float]:
        ms = ctypes.c_double()
        return ms.value if self._lib.nus_upscaler_last_gpu_ms(self._h, ctypes.byref(ms)) == C.OK else None

    @property
    def input_size(self) -> int:
        return self._lib.nus_upscaler_input_size(self._h)

    @property
    def output_size(self) -> int:
        return self._lib.nus_upscaler_output_size(self._h)

    def export_tables(self) -> bytes:
        n = self._lib.nus_upscaler_export_tables(self._h, None, 0)
        if n < 0:
            raise RuntimeError(C.last_error())
        buf = ctypes.create_string_buffer(n)
        if self._lib.nus_upscaler_export_tables(self._h, buf, n) != n:
            raise RuntimeError(C.last_error())
        return buf.raw

    def import_tables(self, blob: bytes) -> None:
        self._check(self._lib.nus_upscaler_import_tables(self._h, blob, len(blob)))


class PyAdvancedWgpuUpscaler(PyWgpuUpscaler):
    """lib.rs:328-729.  The VRAM / strategy methods exist for API parity (the GUI only
    calls them behind `hasattr`); memory is owned by the HIP handle."""

    def __init__(self, quality: str = "quality", algorithm: str = "bilinear", adaptive_quality: bool = True, **kw):
        super().__init__(quality, algorithm, **kw)
        self._adaptive_quality = bool(adaptive_quality)
        self._quality_str = quality.lower() if quality.lower() in _QUALITY else "quality"

    def force_gpu_activation(self) -> None:
        return None

    def set_memory_strategy(self, strategy: str) -> None:
        if str(strategy).lower() not in ("auto", "aggressive", "balanced", "conservative", "minimal"):
            raise ValueError(f"Unknown memory strategy: {strategy}")

    @property
    def adaptive_quality(self) -> bool:
        return self._adaptive_quality

    @adaptive_quality.setter
    def adaptive_quality(self, enabled: bool) -> None:
        self._adaptive_quality = bool(enabled)

    def cleanup_memory(self) -> None:
        return None

    def force_cleanup(self) -> None:
        return None

    def update_gpu_stats(self) -> None:
        return None

    def get_quality_str(self) -> str:
        return self._quality_str

    def set_quality(self, quality: str) -> None:
        q = str(quality).lower()
        if q not in _QUALITY:
            raise ValueError(f"Invalid quality: {quality}")
        self._quality_str = q
        self._check(self._lib.nus_upscaler_set_quality(self._h, _QUALITY[q]))

    def get_vram_stats(self) -> PyVramStats:
        """lib.rs:539-549; the numbers come from hipMemGetInfo of this upscaler's device."""
        free, total = ctypes.c_uint64(), ctypes.c_uint64()
        if self._lib.nus_device_memory_info(self._device, ctypes.byref(free), ctypes.byref(total)) != C.OK:
            raise RuntimeError("No GPU resources available")  # lib.rs:545-547
        mb = 1024.0 * 1024.0
        app = (self.input_size + self.output_size) * 3 / mb  # the three pipeline slots of this handle
        return PyVramStats(total.value / mb, (total.value - free.value) / mb, free.value / mb, app)

    def get_vram_usage_percent(self) -> float:
        """lib.rs:571-584."""
        return self.get_vram_stats().usage_percent

    def get_gpu_info(self) -> dict:
        return {"name": "AMD Instinct MI355X (HIP)", "vendor": "AMD", "backend": "HIP/gfx950",
                "devices": C.device_count()}


class PyVramStats:
    """gpu/memory.rs:731-764: total_mb, used_mb, free_mb, app_allocated_mb, usage_percent."""

    def __init__(self, total_mb: float, used_mb: float, free_mb: float, app_allocated_mb: float):
        self.total_mb, self.used_mb, self.free_mb, self.app_allocated_mb = total_mb, used_mb, free_mb, app_allocated_mb
        self.usage_percent = used_mb / total_mb * 100.0 if total_mb > 0.0 else 0.0

    def __repr__(self) -> str:
        return (f"PyVramStats(total_mb={self.total_mb:.0f}, used_mb={self.used_mb:.0f}, free_mb={self.free_mb:.0f}, "
                f"usage_percent={self.usage_percent:.1f})")


def create_advanced_upscaler(quality: str) -> PyAdvancedWgpuUpscaler:
    """lib.rs:738-741."""
    return PyAdvancedWgpuUpscaler(quality, "bilinear", True)
