"""Optical-flow front end of the frame interpolator ("next" row).

Mirrors the reference's `WgpuFrameInterpolator::build_pyramid` / `compute_coarse_flow`
(nu_scaler_core/src/wgpu_interpolator.rs:969-1203): Gaussian pyramid (5-tap blur H/V + 2x box
downsample) of both frames, Horn-Schunck Jacobi steps at the coarsest level, then -- where the
reference's refine path is dead code -- a coarse-to-fine warm start: x2 bilinear flow upsample
and more Jacobi steps per finer level.  The result is the dense (dx, dy) field that
`WgpuFrameInterpolator.interpolate_py(..., flow=...)` consumes.
"""
from __future__ import annotations

import numpy as np

from . import _capi as C


class FlowEstimator:
    def __init__(self, levels: int = 3, coarse_iterations: int = 50, refine_iterations: int = 10,
                 lambda_: float = 0.02 ** 2, *, device: int = 0):
        # lambda default: the value of the reference's own (ignored) test, wgpu_interpolator.rs:1535
        self._lib = C.lib()
        self._h = self._lib.nus_flow_create()
        if not self._h:
            raise RuntimeError(C.last_error())
        self.levels, self.coarse_iterations = int(levels), int(coarse_iterations)
        self.refine_iterations, self.lambda_ = int(refine_iterations), float(lambda_)
        self._check(self._lib.nus_flow_set_device(self._h, int(device)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.nus_flow_destroy(h)

    def set_tiled(self, enabled) -> None:
        """True / 1 (default): multi-step Horn-Schunck kernels (2: always LDS tiles, 3: always the
        register-pipelined kernel); False / 0: one plain kernel per step.  Same bits either way."""
        self._check(self._lib.nus_flow_set_tiled(self._h, int(enabled)))

    def set_mode(self, mode: str) -> None:
        """"exact" (default): every stage bit-identical to the oracle; "fast": the estimators' Jacobi steps in separable sums,
        reciprocals and FMAs (flow within 1e-3 px; nus_flow_set_mode)."""
        m = {"exact": 0, "fast": 1}.get(str(mode).lower())
        if m is None:
            raise ValueError("mode must be 'exact' or 'fast'")
        self._check(self._lib.nus_flow_set_mode(self._h, m))

    @property
    def mode(self) -> str:
        return "fast" if self._lib.nus_flow_mode(self._h) == 1 else "exact"

    def _check(self, status: int) -> None:
        if status != C.OK:
            raise RuntimeError(self._lib.nus_flow_last_error(self._h).decode("utf-8", "replace"))

    @staticmethod
    def _f32(a, shape=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if shape is not None and a.shape != shape:
            raise ValueError(f"expected shape {shape}, got {a.shape}")
        return a

    # ---- primitives (host arrays in / out) ----------------------------------------
    def rgba8_to_f32(self, img: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        out = np.empty((h, w, 4), np.float32)
        self._check(self._lib.nus_flow_rgba8_to_f32(self._h, img.ctypes.data, w, h, out.ctypes.data))
        return out

    def blur(self, img: np.ndarray) -> np.ndarray:
        img = self._f32(img)
        h, w = img.shape[:2]
        out = np.empty_like(img)
        self._check(self._lib.nus_flow_blur(self._h, img.ctypes.data, w, h, out.ctypes.data))
        return out

    def downsample(self, img: np.ndarray) -> np.ndarray:
        img = self._f32(img)
        h, w = img.shape[:2]
        out = np.empty(((h + 1) // 2, (w + 1) // 2, 4), np.float32)
        self._check(self._lib.nus_flow_downsample(self._h, img.ctypes.data, w, h, out.ctypes.data))
        return out

    def horn_schunck(self, i1: np.ndarray, i2: np.ndarray, flow_in=None, iterations: int = 1, lambda_=None) -> np.ndarray:
        i1, i2 = self._f32(i1), self._f32(i2, np.shape(i1))
        h, w = i1.shape[:2]
        fin = None if flow_in is None else self._f32(flow_in, (h, w, 2))
        out = np.empty((h, w, 2), np.float32)
        self._check(self._lib.nus_flow_horn_schunck(self._h, i1.ctypes.data, i2.ctypes.data,
                                                    None if fin is None else fin.ctypes.data, w, h,
                                                    self.lambda_ if lambda_ is None else float(lambda_),
                                                    int(iterations), out.ctypes.data))
        return out

    def upsample(self, flow: np.ndarray, dw: int, dh: int, scale: float = 1.0) -> np.ndarray:
        flow = self._f32(flow)
        sh, sw = flow.shape[:2]
        out = np.empty((dh, dw, 2), np.float32)
        self._check(self._lib.nus_flow_upsample(self._h, flow.ctypes.data, sw, sh, out.ctypes.data, dw, dh, float(scale)))
        return out

    # ---- full estimator ------------------------------------------------------------
    def estimate(self, frame_a, frame_b, width: int, height: int) -> np.ndarray:
        """RGBA8 frames (bytes or arrays) -> (h, w, 2) float32 flow, pixel delta A -> B."""
        a = np.frombuffer(frame_a, np.uint8) if isinstance(frame_a, (bytes, bytearray, memoryview)) else np.ascontiguousarray(frame_a, np.uint8)
        b = np.frombuffer(frame_b, np.uint8) if isinstance(frame_b, (bytes, bytearray, memoryview)) else np.ascontiguousarray(frame_b, np.uint8)
        n = width * height * 4
        if a.size != n or b.size != n:
            raise ValueError(f"Expected {n} bytes per frame for {width}x{height}x4 RGBA, got frame_a: {a.size} bytes, frame_b: {b.size} bytes")
        out = np.empty((height, width, 2), np.float32)
        self._check(self._lib.nus_flow_estimate(self._h, a.ctypes.data, b.ctypes.data, width, height, self.levels,
                                                self.coarse_iterations, self.refine_iterations, self.lambda_,
                                                out.ctypes.data))
        return out

    def estimate_device(self, d_a: int, d_b: int, width: int, height: int, d_flow_out: int, stream: int = 0) -> None:
        self._check(self._lib.nus_flow_estimate_device(self._h, d_a, d_b, width, height, self.levels,
                                                       self.coarse_iterations, self.refine_iterations, self.lambda_,
                                                       d_flow_out, stream or None))

    def estimate_device_stream(self, d_frames: int, n_frames: int, width: int, height: int, d_flows: int,
                               stream: int = 0) -> None:
        """`n_frames` consecutive RGBA8 frames on the device -> `n_frames - 1` flows (k -> k+1) at `d_flows`;
        same result as one `estimate_device` per pair, each frame's pyramid built once."""
        self._check(self._lib.nus_flow_estimate_device_stream(self._h, d_frames, n_frames, width, height, self.levels,
                                                              self.coarse_iterations, self.refine_iterations,
                                                              self.lambda_, d_flows, stream or None))

    def interpolate_device_stream(self, d_frames: int, n_frames: int, width: int, height: int, time_t: float, d_mid: int,
                                  d_flows: int = 0, stream: int = 0, flow_format: str = "f32") -> None:
        """The reference's intended interpolate() as one pipeline (wgpu_interpolator.rs:881-935: pyramid -> coarse flow -> warp):
        `n_frames` consecutive RGBA8 frames on the device -> the `n_frames - 1` in-between frames at `time_t` at `d_mid`, warped +
        blended with each pair's flow (FMA-mode warp), and the flows themselves at `d_flows` if that is not 0
        (nus_flow_interpolate_device_stream).  flow_format "f16": the flows between estimator and warp (and at `d_flows`) as
        Rg16Float, the reference's live layout -- the f32 flow rounded to nearest even, read as such by the warp."""
        fmt = {"f32": 0, "f16": 1}.get(flow_format)
        if fmt is None:
            raise ValueError("flow_format must be 'f32' or 'f16'")
        self._check(self._lib.nus_flow_interpolate_device_stream(self._h, d_frames, n_frames, width, height, self.levels,
                                                                 self.coarse_iterations, self.refine_iterations, self.lambda_,
                                                                 float(time_t), fmt, d_flows or None, d_mid, stream or None))
