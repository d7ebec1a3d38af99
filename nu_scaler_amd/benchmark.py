"""Python mirror of the reference's benchmark entry points (nu_scaler_core/src/benchmark.rs:70-272):
`py_benchmark_upscaler`, `py_run_comparison_benchmark`, `PyBenchmarkResult`.  Same arguments, same result
fields, same synthetic gradient (benchmark.rs:188-207) and the same per-frame timing of `upscale()` through
the trait-shaped host path (PCIe included) -- the device-resident rate is bench.py's business."""
from __future__ import annotations

import math
import time
from typing import List

from . import _capi as C
from . import synthetic
from .upscaler import PyWgpuUpscaler

_TECH = {"fsr": (C.TECH_FSR, "FSR"), "dlss": (C.TECH_DLSS, "DLSS"), "wgpu": (C.TECH_WGPU, "Wgpu"),
         "fallback": (C.TECH_FALLBACK, "Fallback")}
_QUALITY_DEBUG = {"ultra": "Ultra", "quality": "Quality", "balanced": "Balanced", "performance": "Performance"}


class PyBenchmarkResult:
    """benchmark.rs:24-50 (all fields read-only attributes there)."""

    def __init__(self, **kw):
        self.upscaler_name: str = kw["upscaler_name"]
        self.technology: str = kw["technology"]
        self.quality: str = kw["quality"]
        self.input_width: int = kw["input_width"]
        self.input_height: int = kw["input_height"]
        self.output_width: int = kw["output_width"]
        self.output_height: int = kw["output_height"]
        self.scale_factor: float = kw["scale_factor"]
        self.avg_frame_time_ms: float = kw["avg_frame_time_ms"]
        self.fps: float = kw["fps"]
        self.frames_processed: int = kw["frames_processed"]
        self.total_duration_ms: float = kw["total_duration_ms"]

    def __repr__(self) -> str:
        return (f"PyBenchmarkResult({self.upscaler_name}, {self.technology}/{self.quality}, "
                f"{self.input_width}x{self.input_height}->{self.output_width}x{self.output_height}, "
                f"{self.avg_frame_time_ms:.3f} ms/frame, {self.fps:.1f} fps)")


def _round_half_away(v: float) -> int:
    """Rust f32::round."""
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))


def _benchmark(tech_key: str, quality_key: str, input_width: int, input_height: int, scale_factor: float,
               frame_count: int, test_data: bytes) -> PyBenchmarkResult:
    if len(test_data) < input_width * input_height * 4:  # benchmark.rs:82-86
        raise RuntimeError("Benchmark error: Test data too small for the specified input resolution")
    # UpscalerFactory::create_upscaler (upscale/mod.rs:95-117): Wgpu -> bilinear, everything else -> nearest
    algorithm = "bilinear" if tech_key == "wgpu" else "nearest"
    up = PyWgpuUpscaler(quality_key, algorithm)
    output_width = _round_half_away(input_width * scale_factor)  # benchmark.rs:101-102
    output_height = _round_half_away(input_height * scale_factor)
    try:
        up.initialize(input_width, input_height, output_width, output_height)
        frame = test_data[:input_width * input_height * 4]
        times = []
        t_start = time.perf_counter()
        for _ in range(frame_count):
            t0 = time.perf_counter()
            up.upscale(frame)
            times.append((time.perf_counter() - t0) * 1e3)
        total_ms = (time.perf_counter() - t_start) * 1e3
    except RuntimeError as e:
        raise RuntimeError(f"Benchmark error: {e}") from e
    avg = sum(times) / len(times) if times else float("nan")
    return PyBenchmarkResult(
        upscaler_name=up.name, technology=_TECH[tech_key][1], quality=_QUALITY_DEBUG[quality_key],
        input_width=input_width, input_height=input_height, output_width=output_width, output_height=output_height,
        scale_factor=float(scale_factor), avg_frame_time_ms=avg, fps=1000.0 / avg if times else float("nan"),
        frames_processed=frame_count, total_duration_ms=total_ms)


def py_benchmark_upscaler(technology: str, quality: str, input_width: int, input_height: int, scale_factor: float,
                          frame_count: int) -> PyBenchmarkResult:
    """benchmark.rs:209-254: unknown technology -> fallback, unknown quality -> quality."""
    tech_key = str(technology).lower() if str(technology).lower() in _TECH else "fallback"
    quality_key = str(quality).lower() if str(quality).lower() in _QUALITY_DEBUG else "quality"
    data = synthetic.gradient_frame(input_width, input_height).tobytes()
    return _benchmark(tech_key, quality_key, input_width, input_height, scale_factor, frame_count, data)


def py_run_comparison_benchmark(input_width: int, input_height: int, scale_factor: float,
                                frame_count: int) -> List[PyBenchmarkResult]:
    """benchmark.rs:141-185: every technology x quality; combinations that fail are reported and skipped."""
    data = synthetic.gradient_frame(input_width, input_height).tobytes()
    results = []
    for tech_key in ("fsr", "dlss", "wgpu", "fallback"):
        for quality_key in ("ultra", "quality", "balanced", "performance"):
            try:
                results.append(_benchmark(tech_key, quality_key, input_width, input_height, scale_factor, frame_count, data))
            except RuntimeError as e:
                print(f"Error benchmarking {_TECH[tech_key][1]}/{_QUALITY_DEBUG[quality_key]}: {e}")
    return results
