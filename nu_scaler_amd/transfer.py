"""HBM <-> host for callers of the `*_device` entry points: `nus_download` / `nus_upload` (include/nuscaler_hip.h).

The reference's `upscale()` always hands back host bytes (map the staging buffer, wait, `to_vec`:
nu_scaler_core/src/upscale/mod.rs:1041-1057).  The device-resident API leaves the fetch to the caller; this is the fetch.
The host buffer may be pageable -- it is never handed to the HIP runtime, whose pageable copies pin the caller's pages on the
fly and cache that registration by address (docs/d2h_fault_analysis.md): the bytes travel through pinned chunks the library
owns and are moved by its copy threads.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _capi


def _host_address(buf, writable: bool):
    """(address, nbytes, keepalive) of a C-contiguous buffer."""
    if isinstance(buf, np.ndarray):
        if not buf.flags["C_CONTIGUOUS"] or (writable and not buf.flags["WRITEABLE"]):
            raise TypeError("transfer: needs a C-contiguous" + (" writable" if writable else "") + " array")
        return buf.ctypes.data, buf.nbytes, buf
    mv = memoryview(buf)
    if not mv.c_contiguous or (writable and mv.readonly):
        raise TypeError("transfer: needs a C-contiguous" + (" writable" if writable else "") + " buffer")
    if mv.readonly:  # bytes: ctypes cannot take from_buffer of a read-only object; numpy can
        a = np.frombuffer(mv, np.uint8)
        return a.ctypes.data, a.nbytes, a
    a = (ctypes.c_ubyte * mv.nbytes).from_buffer(mv.cast("B"))
    return ctypes.addressof(a), mv.nbytes, a


def download(device_ptr: int, nbytes: int, out=None, stream: int = 0):
    """`nbytes` from device address `device_ptr` into `out` (a writable buffer of at least that size; default: a fresh
    `bytearray`), ordered after the work already enqueued on `stream` (a hipStream_t as an int; 0 = the null stream).
    Returns `out`."""
    if out is None:
        out = bytearray(nbytes)
    addr, cap, keep = _host_address(out, True)
    if cap < nbytes:
        raise ValueError(f"download: output buffer holds {cap} bytes, {nbytes} needed")
    if _capi.lib().nus_download(addr, device_ptr, nbytes, stream) != _capi.OK:
        raise RuntimeError(_capi.last_error())
    del keep
    return out


def upload(device_ptr: int, data, stream: int = 0) -> None:
    """The bytes of `data` (any C-contiguous buffer) to device address `device_ptr`; `data` may be re-used on return, the
    device bytes are in place for work enqueued on `stream` afterwards."""
    addr, n, keep = _host_address(data, False)
    if _capi.lib().nus_upload(device_ptr, addr, n, stream) != _capi.OK:
        raise RuntimeError(_capi.last_error())
    del keep


_NP_OF_TORCH = {"torch.uint8": np.uint8, "torch.int8": np.int8, "torch.int16": np.int16, "torch.int32": np.int32,
                "torch.int64": np.int64, "torch.float16": np.float16, "torch.float32": np.float32, "torch.float64": np.float64,
                "torch.bool": np.bool_}


def to_numpy(t) -> np.ndarray:
    """A numpy array with the contents of the torch tensor `t`.  A CUDA tensor comes down through `nus_download` on torch's
    current stream of its device (so after everything enqueued there); a CPU tensor is returned as `t.numpy()`."""
    import torch

    if not t.is_cuda:
        return t.numpy()
    t = t.contiguous()
    out = np.empty(tuple(t.shape), _NP_OF_TORCH[str(t.dtype)])
    if out.nbytes:
        download(t.data_ptr(), out.nbytes, out, torch.cuda.current_stream(t.device).cuda_stream)
    return out


def to_device(a, device="cuda:0"):
    """A torch tensor on `device` with the contents of the numpy array (or CPU tensor) `a`, sent through `nus_upload` on
    torch's current stream of that device."""
    import torch

    if isinstance(a, torch.Tensor):
        a = a.numpy()
    a = np.ascontiguousarray(a)
    dt = {v: k for k, v in _NP_OF_TORCH.items()}[a.dtype.type]
    t = torch.empty(a.shape, dtype=getattr(torch, dt.split(".")[1]), device=device)
    if a.nbytes:
        upload(t.data_ptr(), a, torch.cuda.current_stream(t.device).cuda_stream)
    return t
