"""Frame-queue stream: the unit of work of the 1080p -> 4K pipeline and its sharding.

One *unit* = one source frame k of a stream: interpolate (k, k+1) at t to get the
in-between frame, then upscale both frame k and the in-between frame (the GUI flow at
nu_scaler_py/nu_scaler/main.py:999-1008, 1087-1088 upscales the blended frame; the
BASELINE metric "x2 upscale + interp" upscales the real and the inserted frame).

Frames are independent units, so a stream shards across GPUs with no data-path
collective: rank r takes a contiguous chunk of source frames plus one overlap frame so
every (k, k+1) pair is local.  The only cross-GPU traffic is the one-off broadcast of
the filter tables (RCCL over xGMI when the backend is nccl).
"""
from __future__ import annotations

from typing import Tuple

from .interpolator import WgpuFrameInterpolator
from .upscaler import PyWgpuUpscaler


def shard_frames(n_units: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous chunk [start, start+count) of source-frame units owned by `rank`.
    Chunks differ by at most one unit; the caller also needs frame start+count (the
    overlap frame) as the second half of its last pair."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, rem = divmod(int(n_units), world_size)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def build_tables_blob(in_w: int, in_h: int, out_w: int, out_h: int, wgsl_bilinear: bool = False,
                      algorithm: str = "lanczos3") -> bytes:
    """Host-only: the table blob an upscaler of these dimensions and algorithm exports (no GPU needed)."""
    import ctypes

    from . import _capi as C
    from .upscaler import _ALGORITHM

    L = C.lib()
    alg = _ALGORITHM[algorithm]
    n = L.nus_tables_build_blob_for(alg, in_w, in_h, out_w, out_h, int(wgsl_bilinear), None, 0)
    if n < 0:
        raise RuntimeError(C.last_error())
    buf = ctypes.create_string_buffer(n)
    if L.nus_tables_build_blob_for(alg, in_w, in_h, out_w, out_h, int(wgsl_bilinear), buf, n) != n:
        raise RuntimeError(C.last_error())
    return buf.raw


def validate_tables_blob(blob: bytes, in_w: int, in_h: int, out_w: int, out_h: int) -> None:
    from . import _capi as C

    if C.lib().nus_tables_validate_blob(blob, len(blob), in_w, in_h, out_w, out_h) != C.OK:
        raise ValueError(C.last_error())


def broadcast_blob(blob, src: int = 0, device=None, force: bool = False) -> bytes:
    """Broadcast a byte blob from rank `src` (RCCL over xGMI with the nccl backend and a
    cuda `device`; gloo on CPU).  Returns the blob on every rank.  A world of one returns the blob
    untouched unless `force` is set: then the two broadcasts are issued on the one-rank communicator
    all the same (tests/test_rccl_one_rank.py drives RCCL this way on a 1-GPU box)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return bytes(blob)
    if dist.get_world_size() == 1 and not force:
        return bytes(blob)
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    size = int(n.item())
    if rank == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        t = t.to(device) if device is not None else t
    else:
        t = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(t, src)
    return t.cpu().numpy().tobytes()


def broadcast_tables(upscaler: PyWgpuUpscaler, src: int = 0, device=None, force: bool = False) -> int:
    """Broadcast rank `src`'s filter / index tables to every rank so all GPUs use
    bit-identical weights.  Returns the blob size (0 when not distributed).  `force`: also on a
    world of one, and the source rank re-imports what came back through the collective."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 0
    if dist.get_world_size() == 1 and not force:
        return 0
    rank = dist.get_rank()
    blob = broadcast_blob(upscaler.export_tables() if rank == src else b"", src, device, force)
    if rank != src or force:
        upscaler.import_tables(blob)
    return len(blob)


class FramePipeline:
    """Device-resident pipeline over a batch of source frames already in HBM.

    frames : (n_units + 1, h, w, 4) uint8 tensor (the +1 is the overlap frame)
    mid    : (n_units, h, w, 4)          in-between frames
    up_real, up_mid : (n_units, oh, ow, 4) upscaled real / in-between frames
    """

    def __init__(self, width: int, height: int, scale: int = 2, algorithm: str = "lanczos3", time_t: float = 0.5,
                 device: int = 0, lanczos_mode: str = "fma"):
        self.w, self.h = int(width), int(height)
        self.ow, self.oh = self.w * scale, self.h * scale
        self.t = float(time_t)
        self.upscaler = PyWgpuUpscaler("quality", algorithm, device=device, lanczos_mode=lanczos_mode)
        self.upscaler.initialize(self.w, self.h, self.ow, self.oh)
        self.interp = WgpuFrameInterpolator(device=device)
        self.frame_bytes = self.w * self.h * 4
        self._aux = None

    # algorithmic bytes / pixels of one unit
    @property
    def unit_bytes(self) -> int:
        up = self.frame_bytes + self.ow * self.oh * 4
        return 3 * self.frame_bytes + 2 * up

    @property
    def unit_pixels(self) -> int:
        return 3 * self.w * self.h + 2 * (self.w * self.h + self.ow * self.oh)

    def alloc(self, n_units: int, device):
        import torch

        mid = torch.empty((n_units, self.h, self.w, 4), dtype=torch.uint8, device=device)
        up_real = torch.empty((n_units, self.oh, self.ow, 4), dtype=torch.uint8, device=device)
        up_mid = torch.empty((n_units, self.oh, self.ow, 4), dtype=torch.uint8, device=device)
        return mid, up_real, up_mid

    def step(self, frames, mid, up_real, up_mid, stream: int = 0) -> None:
        """Enqueue one pass over the batch on `stream` (a hipStream_t as int; 0 = default)."""
        n = mid.shape[0]
        base = frames.data_ptr()
        fb = self.frame_bytes
        self.interp.interpolate_device(base, fb, base + fb, fb, 0, self.w, self.h, self.t, mid.data_ptr(), n, stream)
        self.upscaler.upscale_device(base, up_real.data_ptr(), n, stream)
        self.upscaler.upscale_device(mid.data_ptr(), up_mid.data_ptr(), n, stream)

    def step_fused(self, frames, up_real, up_mid, stream: int = 0) -> None:
        """Same outputs without materialising the in-between frames: the blend happens inside the
        second upscale's row loads (two launches instead of three, 16.6 MB less HBM traffic per unit)."""
        n = up_real.shape[0]
        base = frames.data_ptr()
        fb = self.frame_bytes
        self.upscaler.upscale_device(base, up_real.data_ptr(), n, stream)
        self.upscaler.upscale_blend_device(base, fb, base + fb, fb, self.t, up_mid.data_ptr(), n, stream)

    def step_unit(self, frames, mid, up_real, up_mid, stream: int = 0) -> None:
        """The same three outputs as `step` from ONE launch of the x2 resize kernel (plus the two edge-column passes): two waves
        per (frame, row block, strip) side by side, one up-scaling frame k, one blending frames k and k+1 on load, storing
        the in-between rows and up-scaling them (nus_upscaler_upscale_unit_device).  Bit-identical to `step`."""
        n = up_real.shape[0]
        base = frames.data_ptr()
        fb = self.frame_bytes
        self.upscaler.upscale_unit_device(base, fb, base + fb, fb, self.t, 0 if mid is None else mid.data_ptr(),
                                          up_real.data_ptr(), up_mid.data_ptr(), n, stream)

    def step_motion(self, frames, flows, mid, up_real, up_mid, stream: int = 0, levels: int = 3,
                    coarse_iterations: int = 50, refine_iterations: int = 10, flow_mode: str = "exact") -> None:
        """Motion-compensated variant of `step`: a dense flow per pair from the
        pyramid + Horn-Schunck front end into `flows` ((n_units, h, w, 2) float32), then warp + blend with it
        instead of the reference's zero flow.  The flows are estimated pair by pair (several launches per pair,
        each frame's pyramid built once), the warp and both upscales run once over the whole batch."""
        from .flow import FlowEstimator

        if getattr(self, "_flow", None) is None:
            self._flow = FlowEstimator(levels, coarse_iterations, refine_iterations, device=self.upscaler._device)
        if self._flow.mode != flow_mode:
            self._flow.set_mode(flow_mode)  # "fast": the Jacobi steps in separable sums / reciprocals / FMAs (flow within 1e-3 px)
        n = mid.shape[0]
        base = frames.data_ptr()
        fb = self.frame_bytes
        self._flow.estimate_device_stream(base, n + 1, self.w, self.h, flows.data_ptr(), stream)
        self.interp.interpolate_device(base, fb, base + fb, fb, flows.data_ptr(), self.w, self.h, self.t, mid.data_ptr(), n,
                                       stream)
        self.upscaler.upscale_device(base, up_real.data_ptr(), n, stream)
        self.upscaler.upscale_device(mid.data_ptr(), up_mid.data_ptr(), n, stream)

    def step_overlapped(self, frames, mid, up_real, up_mid) -> None:
        """Same work with the blend on a second stream: the Lanczos kernel is bound by SIMD
        time (VALU issue + store feed) and leaves HBM bandwidth idle, the zero-flow blend is
        purely bandwidth bound, and the upscale of the real frames does not depend on the
        blend -- so the two run concurrently and the blend is (mostly) hidden.
        Uses torch's current stream as the main stream."""
        import torch

        if self._aux is None:
            self._aux = torch.cuda.Stream()
            self._ev_start = torch.cuda.Event()
            self._ev_mid = torch.cuda.Event()
        main = torch.cuda.current_stream()
        n = mid.shape[0]
        base = frames.data_ptr()
        fb = self.frame_bytes
        self._ev_start.record(main)
        self._aux.wait_event(self._ev_start)  # the previous step's readers of `mid` are done
        self.interp.interpolate_device(base, fb, base + fb, fb, 0, self.w, self.h, self.t, mid.data_ptr(), n,
                                       self._aux.cuda_stream)
        self._ev_mid.record(self._aux)
        self.upscaler.upscale_device(base, up_real.data_ptr(), n, main.cuda_stream)
        main.wait_event(self._ev_mid)
        self.upscaler.upscale_device(mid.data_ptr(), up_mid.data_ptr(), n, main.cuda_stream)
