"""nu_scaler_amd -- MI355X (gfx950) implementation of nu_scaler_core's per-pixel hot
path (nearest / bilinear / Lanczos-3 upscaling, two-frame warp + blend) behind the
reference's own Python surface:  `import nu_scaler_amd as nu_scaler_core`.

The compute path is libnuscaler_hip.so (hand-written HIP kernels behind the C ABI of
include/nuscaler_hip.h).  There is no CPU fallback: constructing any class without the
library raises, and compute calls without a HIP device raise RuntimeError.

When the same process also uses PyTorch-ROCm (bench.py, the device-resident tests), import torch
BEFORE this package: torch ships its own HIP runtime and fails to see the GPU if /opt/rocm's
libamdhip64 (pulled in by libnuscaler_hip.so) is loaded first.
"""
from . import _capi, transfer
from ._capi import NuScalerLibraryError, PinnedBuffer, build, device_count
from .benchmark import PyBenchmarkResult, py_benchmark_upscaler, py_run_comparison_benchmark
from .flow import FlowEstimator
from .imagefile import interpolate_image_files, upscale_image_file
from .interpolator import WgpuFrameInterpolator
from .queue import FrameBuffer, swizzle_bgra_to_rgba_device
from .launch import launch_ranks
from .stream import (FramePipeline, ShardedStream, SyntheticSource, broadcast_blob, broadcast_tables, build_tables_blob,
                     gather_rows, run_sharded, shard_frames, spread, validate_tables_blob)
from .transfer import download, upload
from .upscaler import PyAdvancedWgpuUpscaler, PyVramStats, PyWgpuUpscaler, create_advanced_upscaler

# module constants of the reference's #[pymodule] (nu_scaler_core/src/lib.rs:746-761)
QUALITY_ULTRA = _capi.QUALITY_ULTRA
QUALITY_QUALITY = _capi.QUALITY_QUALITY
QUALITY_BALANCED = _capi.QUALITY_BALANCED
QUALITY_PERFORMANCE = _capi.QUALITY_PERFORMANCE
TECH_FSR = _capi.TECH_FSR
TECH_DLSS = _capi.TECH_DLSS
TECH_WGPU = _capi.TECH_WGPU
TECH_FALLBACK = _capi.TECH_FALLBACK
VENDOR_NVIDIA, VENDOR_AMD, VENDOR_INTEL, VENDOR_OTHER = 0, 1, 2, 3


def install_fatal_trace(fd: int = 2) -> bool:
    """nus_install_fatal_trace: a process that dies of a fatal signal in native code writes the native backtrace of the raising
    thread, the host ranges the library holds and /proc/self/maps to (a duplicate of) `fd` first.  Off unless called."""
    return _capi.lib().nus_install_fatal_trace(int(fd)) == _capi.OK


def create_fsr_upscaler(_quality: str):
    """lib.rs:791-806: FSR3 is not part of this build either."""
    raise NotImplementedError("FSR3 support is not enabled in this build.")


__all__ = [
    "PyWgpuUpscaler", "PyAdvancedWgpuUpscaler", "PyVramStats", "create_advanced_upscaler", "create_fsr_upscaler",
    "upscale_image_file", "interpolate_image_files",
    "PyBenchmarkResult", "py_benchmark_upscaler", "py_run_comparison_benchmark",
    "WgpuFrameInterpolator", "FlowEstimator", "FrameBuffer", "swizzle_bgra_to_rgba_device", "FramePipeline", "shard_frames", "broadcast_tables",
    "ShardedStream", "SyntheticSource", "run_sharded", "gather_rows", "spread", "launch_ranks",
    "broadcast_blob", "build_tables_blob", "validate_tables_blob",
    "NuScalerLibraryError", "PinnedBuffer", "build", "device_count", "download", "upload", "transfer", "install_fatal_trace",
    "QUALITY_ULTRA", "QUALITY_QUALITY", "QUALITY_BALANCED", "QUALITY_PERFORMANCE",
    "TECH_FSR", "TECH_DLSS", "TECH_WGPU", "TECH_FALLBACK",
    "VENDOR_NVIDIA", "VENDOR_AMD", "VENDOR_INTEL", "VENDOR_OTHER",
]
