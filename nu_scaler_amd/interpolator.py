"""Python mirror of the reference's `WgpuFrameInterpolator` pyclass
(nu_scaler_core/src/wgpu_interpolator.rs:130-498), bound to the HIP C ABI."""
from __future__ import annotations

import ctypes
from typing import Optional

from . import _capi as C
from .upscaler import _as_buffer, _out_buffer

_PRESETS = {  # wgpu_interpolator.rs:117-125
    "8x8": C.WG_SQUARE_8X8, "square8x8": C.WG_SQUARE_8X8,
    "16x16": C.WG_SQUARE_16X16, "square16x16": C.WG_SQUARE_16X16,
    "32x8": C.WG_WIDE_32X8, "wide32x8": C.WG_WIDE_32X8, "wide": C.WG_WIDE_32X8,
    "8x32": C.WG_TALL_8X32, "tall8x32": C.WG_TALL_8X32, "tall": C.WG_TALL_8X32,
}


class WgpuFrameInterpolator:
    """`WgpuFrameInterpolator(workgroup_preset_str=None)` (wgpu_interpolator.rs:172-174).
    Unknown / missing presets default to Wide32x8 as in the reference; the preset is
    recorded only -- the HIP kernels use their own wave64 launch shape (and, unlike the
    reference's dispatch at :372-374, always cover the whole frame)."""

    def __init__(self, workgroup_preset_str: Optional[str] = None, *, device: int = 0):
        self._lib = C.lib()
        preset = _PRESETS.get(str(workgroup_preset_str).lower(), C.WG_WIDE_32X8) if workgroup_preset_str else C.WG_WIDE_32X8
        self._h = self._lib.nus_interp_create(preset)
        if not self._h:
            raise RuntimeError(C.last_error())
        if self._lib.nus_interp_set_device(self._h, int(device)) != C.OK:
            raise RuntimeError(self._err())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.nus_interp_destroy(h)

    def _err(self) -> str:
        return self._lib.nus_interp_last_error(self._h).decode("utf-8", "replace")

    def _raise(self, status: int) -> None:
        if status == C.OK:
            return
        msg = self._err()
        if status == C.ERR_SIZE_MISMATCH:  # PyValueError at wgpu_interpolator.rs:233-238
            raise ValueError(msg)
        raise RuntimeError(msg)

    def interpolate_py(self, frame_a_bytes, frame_b_bytes, width: int, height: int, *, time_t: float = 0.5,
                       flow=None) -> bytes:
        """wgpu_interpolator.rs:215-225.  `flow` (optional, not in the reference's
        signature): (h, w, 2) float32 buffer of pixel deltas A->B; None = zero flow,
        which is what the reference always uses."""
        a_addr, a_len, ka = _as_buffer(frame_a_bytes)
        b_addr, b_len, kb = _as_buffer(frame_b_bytes)
        f_addr, kf = None, None
        expected = int(width) * int(height) * 4
        if flow is not None:
            f_addr, f_len, kf = _as_buffer(flow)
            if f_len != expected * 2:
                raise ValueError(f"Expected {expected * 2} bytes of flow for {width}x{height}x2 f32, got {f_len}")
        out, oarr, oaddr = _out_buffer(expected)
        st = self._lib.nus_interp_interpolate(self._h, a_addr, a_len, b_addr, b_len, f_addr, width, height,
                                              float(time_t), oaddr, expected)
        del oarr, ka, kb, kf
        self._raise(st)
        return out

    # -- trait FrameInterpolator (nu_scaler_core/src/interpolation/mod.rs:29-44; dead code in the reference, same shape here)
    def initialize(self, width: int, height: int) -> None:
        self._raise(self._lib.nus_interp_initialize(self._h, int(width), int(height)))

    def interpolate(self, frame1, frame2, t: float) -> bytes:
        """Frames of the initialize() size, zero flow; RuntimeError("Interpolator not initialized") before it."""
        a_addr, a_len, ka = _as_buffer(frame1)
        b_addr, b_len, kb = _as_buffer(frame2)
        out, oarr, oaddr = _out_buffer(a_len)
        st = self._lib.nus_interp_interpolate_frames(self._h, a_addr, a_len, b_addr, b_len, float(t), oaddr, a_len)
        del oarr, ka, kb
        self._raise(st)
        return out

    @property
    def name(self) -> str:
        return self._lib.nus_interp_name(self._h).decode()

    _QUALITY = {"high": 0, "medium": 1, "low": 2}

    def set_quality(self, quality) -> None:
        q = self._QUALITY.get(str(quality).lower(), quality)
        self._raise(self._lib.nus_interp_set_quality(self._h, int(q)))

    @property
    def quality(self) -> str:
        return ("high", "medium", "low")[self._lib.nus_interp_quality(self._h)]

    def interpolate_device(self, d_a: int, a_stride: int, d_b: int, b_stride: int, d_flow: int, width: int,
                           height: int, time_t: float, d_out: int, n_pairs: int = 1, stream: int = 0) -> None:
        self._raise(self._lib.nus_interp_interpolate_device(self._h, d_a, a_stride, d_b, b_stride, d_flow or None,
                                                            width, height, float(time_t), d_out, n_pairs, stream or None))

    def set_input_format(self, fmt: str) -> None:
        """"rgba" (default) or "bgra" for both input frames; the new frame is RGBA."""
        f = {"rgba": C.FORMAT_RGBA8, "bgra": C.FORMAT_BGRA8, "rgbx": C.FORMAT_RGBX8, "bgrx": C.FORMAT_BGRX8}.get(str(fmt).lower())
        if f is None:
            raise ValueError("input format must be 'rgba', 'bgra', 'rgbx' or 'bgrx'")
        self._raise(self._lib.nus_interp_set_input_format(self._h, f))

    def set_mode(self, mode: str) -> None:
        """Arithmetic of the dense-flow warp: "exact" (default; the CPU's roundings, bit-exact against the oracle) or "fma"
        (fused lerps, within 1 LSB).  The zero-flow blend is exact either way."""
        m = {"exact": 0, "fma": 1}.get(str(mode).lower())
        if m is None:
            raise ValueError("mode must be 'exact' or 'fma'")
        self._raise(self._lib.nus_interp_set_mode(self._h, m))

    @property
    def mode(self) -> str:
        return "fma" if self._lib.nus_interp_mode(self._h) == 1 else "exact"

    def set_flow_format(self, fmt: str) -> None:
        """Element type of the device flow field `interpolate_device` reads: "f32" (default, 2 x f32 per pixel) or
        "f16" (2 x half per pixel: the Rg16Float texture of wgpu_interpolator.rs:276, half the bytes)."""
        f = {"f32": 0, "f16": 1}.get(str(fmt).lower())
        if f is None:
            raise ValueError("flow format must be 'f32' or 'f16'")
        self._raise(self._lib.nus_interp_set_flow_format(self._h, f))

    def get_last_gpu_duration_ms(self) -> Optional[float]:
        """wgpu_interpolator.rs:494-497: None until an interpolation has run."""
        ms = ctypes.c_double()
        return ms.value if self._lib.nus_interp_last_gpu_ms(self._h, ctypes.byref(ms)) == C.OK else None
