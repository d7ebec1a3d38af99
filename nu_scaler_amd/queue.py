"""Bounded frame queue ("next" row): the legacy
`FrameBuffer` (Nu_scale/src/capture/frame_buffer.rs:11-100) -- capacity 5 by default
(Nu_scale/src/lib.rs), drop-oldest on overflow, consumers take the latest frame."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

from . import _capi as C
from .upscaler import _as_buffer


class FrameBuffer:
    def __init__(self, capacity: int = 5, max_frame_bytes: int = 3840 * 2160 * 4):
        self._lib = C.lib()
        self._h = self._lib.nus_frame_queue_create(int(capacity))
        if not self._h:
            raise RuntimeError(C.last_error())
        self._max = int(max_frame_bytes)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.nus_frame_queue_destroy(h)

    def add_frame(self, frame, width: int, height: int) -> int:
        """Copy a frame in; when full the oldest frame is dropped.  Returns frames dropped so far."""
        addr, n, keep = _as_buffer(frame)
        if n != width * height * 4:
            raise ValueError(f"Expected {width * height * 4} bytes for {width}x{height}x4 RGBA, got {n}")
        r = self._lib.nus_frame_queue_add(self._h, addr, width, height)
        del keep
        if r < 0:
            raise RuntimeError(C.last_error())
        return int(r)

    def _take(self, fn, timeout_ms: int) -> Optional[Tuple[bytes, int, int, int]]:
        buf = ctypes.create_string_buffer(self._max)
        w, h, seq = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        r = fn(self._h, int(timeout_ms), buf, self._max, ctypes.byref(w), ctypes.byref(h), ctypes.byref(seq))
        if r < 0:
            raise RuntimeError(C.last_error())
        if r == 0:
            return None
        return buf.raw[: w.value * h.value * 4], w.value, h.value, seq.value

    def get_latest_frame(self, timeout_ms: int = 0):
        """(bytes, width, height, sequence) of the newest frame, or None (get_latest_frame[_timeout])."""
        return self._take(self._lib.nus_frame_queue_latest, timeout_ms)

    def pop_frame(self, timeout_ms: int = 0):
        """Oldest frame, removed from the queue, or None."""
        return self._take(self._lib.nus_frame_queue_pop, timeout_ms)

    def __len__(self) -> int:
        return int(self._lib.nus_frame_queue_size(self._h))

    @property
    def capacity(self) -> int:
        return int(self._lib.nus_frame_queue_capacity(self._h))

    @property
    def dropped(self) -> int:
        return int(self._lib.nus_frame_queue_dropped(self._h))


def swizzle_bgra_to_rgba_device(d_in: int, d_out: int, n_pixels: int, stream: int = 0) -> None:
    """BGRA -> RGBA on the GPU (nu_scaler_core/src/lib.rs:251-270 does it on the CPU)."""
    if C.lib().nus_swizzle_bgra_to_rgba_device(d_in, d_out, int(n_pixels), stream or None) != C.OK:
        raise RuntimeError(C.last_error())
