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

from typing import Optional, Tuple

from . import transfer
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
    on_gpu = device is not None and torch.device(device).type == "cuda"
    if rank == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        if on_gpu:  # (host <-> HBM through the library's pinned ring, never the runtime's pageable copy: transfer.py)
            t = transfer.to_device(t, device)
        elif device is not None:
            t = t.to(device)
    else:
        t = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(t, src)
    return transfer.to_numpy(t).tobytes()


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
                    coarse_iterations: int = 50, refine_iterations: int = 10, flow_mode: str = "exact", pipelined: bool = False,
                    chunk: int = 150, fused_warp: bool = False, flow_format: Optional[str] = None) -> None:
        """Motion-compensated variant of `step`: a dense flow per pair from the
        pyramid + Horn-Schunck front end into `flows` ((n_units, h, w, 2) float32), then warp + blend with it
        instead of the reference's zero flow.  The flows are estimated pair by pair (several launches per pair,
        each frame's pyramid built once), the warp and both upscales run once over the whole batch.

        pipelined (round 5): the intended interpolate() of the reference is a PIPELINE -- pyramid -> coarse flow -> warp
        (wgpu_interpolator.rs:881-935) -- and so is this: the batch goes through in chunks of `chunk` units, the estimator of
        chunk i+1 on the caller's stream while warp + both upscales of chunk i run on a second stream.  The estimator's launches
        are bound by memory and latency as often as by instruction issue, the warp and the resize kernels by instruction issue
        alone, so the two streams fill each other's gaps; same kernels, same bytes, bit-identical outputs.  Uses torch for the
        second stream and the events; `stream` must be torch's current stream (0 = that).

        fused_warp (round 5): estimator and warp through ONE entry point (FlowEstimator.interpolate_device_stream); `flows` may then
        be None (they stay in the estimator's workspace).  The warp is the FMA-mode one whatever `self.interp`'s mode is; same bytes
        as the separate FMA-mode warp.  (The in-kernel fusion -- the last Horn-Schunck launch warping with the flow it has just
        finished -- exists behind NUS_HS_FUSED_WARP=1: identical bytes, measured slower.)"""
        from .flow import FlowEstimator

        if getattr(self, "_flow", None) is None:
            self._flow = FlowEstimator(levels, coarse_iterations, refine_iterations, device=self.upscaler._device)
        if self._flow.mode != flow_mode:
            self._flow.set_mode(flow_mode)  # "fast": the Jacobi steps in separable sums / reciprocals / FMAs (flow within 1e-3 px)
        n = mid.shape[0]
        base = frames.data_ptr()
        fb = self.frame_bytes
        if flows is None and not fused_warp:
            raise ValueError("step_motion: flows=None needs fused_warp=True")
        # Between estimator and warp the flow travels as Rg16Float by default in FAST mode -- the layout the reference's live path
        # binds (wgpu_interpolator.rs:275-276), half the hand-off's bytes, 5e-4 px of rounding on a 1-2 px flow inside FAST's own
        # 1e-3 px contract; "f32" stays selectable, and is what the exact mode and a caller who wants the flows stored get.
        if flow_format is None:
            flow_format = "f16" if (fused_warp and flow_mode == "fast" and flows is None) else "f32"
        if flow_format != "f32" and not fused_warp:
            raise ValueError("step_motion: flow_format 'f16' (the Rg16Float hand-off) needs fused_warp=True")
        fl0 = 0 if flows is None else flows.data_ptr()

        def flow_and_warp(k0, m, st):  # units [k0, k0 + m): their flows (if wanted) and in-between frames
            a = base + k0 * fb
            fl = fl0 + k0 * fb * (1 if flow_format == "f16" else 2) if fl0 else 0
            if fused_warp:
                self._flow.interpolate_device_stream(a, m + 1, self.w, self.h, self.t, mid.data_ptr() + k0 * fb, fl, st, flow_format)
                return False
            self._flow.estimate_device_stream(a, m + 1, self.w, self.h, fl, st)
            return True  # the warp is still to do

        if not pipelined or n <= chunk:
            if flow_and_warp(0, n, stream):
                self.interp.interpolate_device(base, fb, base + fb, fb, fl0, self.w, self.h, self.t, mid.data_ptr(), n, stream)
            self.upscaler.upscale_device(base, up_real.data_ptr(), n, stream)
            self.upscaler.upscale_device(mid.data_ptr(), up_mid.data_ptr(), n, stream)
            return
        import torch

        main = torch.cuda.current_stream()
        if stream not in (0, main.cuda_stream):
            raise ValueError("step_motion(pipelined=True) runs on torch's current stream")
        if self._aux is None:
            self._aux = torch.cuda.Stream()
            self._ev_start = torch.cuda.Event()
            self._ev_mid = torch.cuda.Event()
        if getattr(self, "_ev_chunks", None) is None or len(self._ev_chunks) < (n + chunk - 1) // chunk:
            self._ev_chunks = [torch.cuda.Event() for _ in range((n + chunk - 1) // chunk)]
        aux = self._aux
        self._ev_start.record(main)
        aux.wait_event(self._ev_start)  # whatever the caller queued before (readers of the output buffers) is done
        ob = self.ow * self.oh * 4
        # the up-scaling of the REAL frames depends on no flow: all of it goes first on the second stream, beside the estimator of
        # the first chunk (which would otherwise have the GPU to itself while its memory-bound launches leave the SIMDs waiting)
        real_first = getattr(self, "_motion_real_first", True)  # (dev switch for tools/motion_bench.py: 19.7-19.8 -> 19.5 ms per 300 units)
        if real_first:
            self.upscaler.upscale_device(base, up_real.data_ptr(), n, aux.cuda_stream)
        for ci, k0 in enumerate(range(0, n, chunk)):
            m = min(chunk, n - k0)
            warp_to_do = flow_and_warp(k0, m, main.cuda_stream)
            self._ev_chunks[ci].record(main)
            aux.wait_event(self._ev_chunks[ci])
            a = base + k0 * fb
            if warp_to_do:
                self.interp.interpolate_device(a, fb, a + fb, fb, fl0 + k0 * fb * 2, self.w, self.h, self.t,
                                               mid.data_ptr() + k0 * fb, m, aux.cuda_stream)
            if not real_first:
                self.upscaler.upscale_device(a, up_real.data_ptr() + k0 * ob, m, aux.cuda_stream)
            self.upscaler.upscale_device(mid.data_ptr() + k0 * fb, up_mid.data_ptr() + k0 * ob, m, aux.cuda_stream)
        self._ev_mid.record(aux)
        main.wait_event(self._ev_mid)  # the caller's stream sees every output complete

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


# ---- the sharded frame-queue stream: one process per GPU, independent frames, no data-path collective ----------------------
#
# north_star: "a frame-queue stream shards independent frames across the 8 GPUs of one node with RCCL broadcast of shared LUTs
# over xGMI only".  `ShardedStream` is one rank of it -- placement, process group, pipeline, LUT broadcast, this rank's contiguous
# shard of the stream (+ the overlap frame), the step loop, the gather of every rank's numbers -- and `run_sharded` the whole
# job of one rank in one call.  `python -m nu_scaler_amd.cli stream --gpus N` and bench.py are both built on it.


def gather_rows(row, world, dist=None, torch=None, comm_dev=None, force=False):
    """Every rank's row of numbers on every rank: ONE all_gather of a float64 vector (keys sorted; None travels as NaN).
    world == 1: no collective (unless `force`: the collective is issued on the one-rank communicator all the same)."""
    keys = sorted(row)
    if world == 1 and not force:
        return [dict(row)]
    vals = [float("nan") if row[k] is None else float(row[k]) for k in keys]
    t = torch.tensor(vals, dtype=torch.float64, device=comm_dev)
    got = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(got, t)
    rows = []
    for g in got:
        rows.append({k: (None if x != x else x) for k, x in zip(keys, g.cpu().tolist())})
    return rows


def spread(rows, key, digits=4):
    """{min, max, by_rank} of one gathered column (None where no rank has it)."""
    xs = [r.get(key) for r in rows]
    have = [x for x in xs if x is not None]
    if not have:
        return None
    return {"min": round(min(have), digits), "max": round(max(have), digits),
            "by_rank": [None if x is None else round(x, digits) for x in xs]}


class SyntheticSource:
    """Frames [first, first + n) of a deterministic synthetic stream as an (n, h, w, 4) uint8 tensor on `device`, whoever asks
    for them: frame k depends on k alone (never on the shard or the chunking), so a stream cut into shards is the stream.
    pattern "gradient": S1 of SURVEY.md 8(d) (benchmark.rs:188-207) moving 1 px per frame, opaque; "noise": S3-like uniform
    bytes, every frame from its own seed."""

    def __init__(self, pattern: str = "gradient", seed: int = 0x5EED):
        if pattern not in ("gradient", "noise"):
            raise ValueError("pattern must be 'gradient' or 'noise'")
        self.pattern, self.seed = pattern, int(seed)

    def __call__(self, first: int, n: int, width: int, height: int, device):
        import torch

        from . import synthetic as syn

        out = torch.empty((n, height, width, 4), dtype=torch.uint8, device=device)
        if self.pattern == "gradient":
            for c0 in range(0, n, 16):  # in chunks: the int64 temporaries are 8x a frame
                c1 = min(c0 + 16, n)
                out[c0:c1] = syn.gradient_stream_torch(c1 - c0, width, height, device, first=first + c0)
        else:
            for k in range(n):
                out[k] = syn.noise_stream_torch(1, width, height, device, seed=self.seed + first + k)[0]
        return out


class _StdoutToStderr:
    """File descriptor 1 pointed at file descriptor 2 for the duration of a `with`: RCCL prints a version banner (five lines:
    "RCCL version : ...", "Librccl path : ...") on STDOUT when rank 0 creates its communicator, and a job's stdout is for its
    one JSON line."""

    def __enter__(self):
        import os
        import sys

        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import os
        import sys

        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class ShardedStream:
    """One rank of the sharded stream.  Construction does, in this order (the order matters):

      1. rank / world from the launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK, LOCAL_WORLD_SIZE; absent: a world of one);
      2. placement: the process onto the CPUs of its GPU's NUMA node, host threads sized to its share -- BEFORE its first HIP
         call (placement.bind_rank), checked against HIP's own word afterwards (placement.verify_after_init);
      3. the process group (backend "nccl" = RCCL over xGMI; "gloo" for rehearsals on host tensors), unless one exists;
      4. the pipeline on this rank's GPU, and rank 0's filter / index tables broadcast to everyone (the job's only collective
         on the data side, a few KiB once);
      5. this rank's contiguous shard of `total_units` units, frames [start, start + count] from `source` (count + 1 frames: the
         overlap frame closes the last pair), and its output buffers -- everything resident in HBM.

    `run(steps, warmup)` then times `steps` passes over the shard between two barriers; `gather(row)` brings every rank's
    numbers to every rank; `close()` tears down what the object created.  No frame ever crosses to another GPU.

    pipeline_factory(width, height, device_index, self) -> an object with .alloc(n, device), .step_unit / .step(frames, mid,
    up_real, up_mid, stream), .unit_pixels, .unit_bytes and (optionally) .upscaler for the table broadcast; default: FramePipeline
    on the HIP device.  (The CPU tests pass a host-tensor stand-in: the product itself has no CPU path.)"""

    def __init__(self, total_units: int, width: int, height: int, *, source=None, backend: str = "nccl", bind: bool = True,
                 force_device: int = -1, schedule: str = "unit", algorithm: str = "lanczos3", time_t: float = 0.5,
                 lanczos_mode: str = "fma", pipeline_factory=None, environ=None, device_kind: str = "cuda",
                 force_collectives: bool = False, resident: bool = True):
        import os

        env = os.environ if environ is None else environ
        self.world = int(env.get("WORLD_SIZE", "1"))
        self.rank = int(env.get("RANK", "0"))
        self.local_rank = int(env.get("LOCAL_RANK", "0"))
        self.local_world = int(env.get("LOCAL_WORLD_SIZE", str(self.world)))
        if not (0 <= self.rank < self.world):
            raise ValueError(f"RANK {self.rank} outside WORLD_SIZE {self.world}")
        if schedule not in ("unit", "three-stage", "fused"):
            raise ValueError("schedule must be 'unit', 'three-stage' or 'fused'")
        self.width, self.height, self.schedule, self.backend = int(width), int(height), schedule, backend
        self.device_index = int(force_device) if force_device >= 0 else self.local_rank
        on_gpu = device_kind == "cuda"

        # 2. placement, before anything of this process touches HIP
        from . import placement as plc

        slot = self.local_rank if force_device >= 0 else None
        self.placement = plc.bind_rank(self.device_index, self.local_world, apply=bool(bind) and on_gpu, slot=slot)
        if not bind:
            self.placement.update(bound=False, why_not="binding turned off by the caller")

        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        if on_gpu:
            if not torch.cuda.is_available():
                raise RuntimeError("the sharded stream needs a HIP device (no CPU fallback)")
            torch.cuda.set_device(self.device_index)
            self.device = torch.device("cuda", self.device_index)
        else:
            self.device = torch.device("cpu")
        nccl = backend == "nccl"
        if nccl and not on_gpu:
            raise ValueError("backend 'nccl' needs device_kind 'cuda'")
        self.comm_device = self.device if nccl else torch.device("cpu")
        # 3. the process group
        # force_collectives: a world of ONE still creates its communicator and issues every collective of the job on it (the LUT
        # broadcast, the barriers, the gather) -- how a one-GPU box rehearses the RCCL calls of the 8-GPU run
        self._collectives = self.world > 1 or bool(force_collectives)
        self._own_group = False
        if self._collectives and not (dist.is_available() and dist.is_initialized()):
            env_set = os.environ.setdefault
            env_set("MASTER_ADDR", "127.0.0.1")
            env_set("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.world == 1 and "MASTER_PORT" not in os.environ:  # force_collectives outside a launcher: a rendezvous of its own
                from .launch import free_port

                os.environ["MASTER_PORT"] = str(free_port())
            with _StdoutToStderr():  # (RCCL's banner: see the class)
                if nccl:
                    dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
                    dist.barrier(device_ids=[self.device_index])  # the communicator exists (and has said so) before anything is timed
                else:
                    dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            self._own_group = True
        try:
            self._finish_init(plc, torch, on_gpu, bind, pipeline_factory, algorithm, time_t, lanczos_mode, total_units, source,
                              force_collectives, resident)
        except BaseException:
            self.close()  # a rank that fails while it sets up must not leave its process group behind
            raise

    def _finish_init(self, plc, torch, on_gpu, bind, pipeline_factory, algorithm, time_t, lanczos_mode, total_units, source,
                     force_collectives, resident=True):
        hip_bdf = None
        if on_gpu:
            try:
                pr = torch.cuda.get_device_properties(self.device_index)
                hip_bdf = f"{int(getattr(pr, 'pci_domain_id', 0)):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0"
            except Exception as e:  # an older torch without the PCI fields
                self.placement["gpu_bdf_by_hip"] = f"unavailable: {type(e).__name__}"
        plc.verify_after_init(self.placement, hip_bdf, apply=bool(bind) and on_gpu)
        # 4. pipeline + shared LUTs
        if pipeline_factory is None:
            self.pipeline = FramePipeline(self.width, self.height, 2, algorithm, time_t, device=self.device_index,
                                          lanczos_mode=lanczos_mode)
        else:
            self.pipeline = pipeline_factory(self.width, self.height, self.device_index, self)
        up = getattr(self.pipeline, "upscaler", None)
        self.lut_bytes = broadcast_tables(up, 0, self.comm_device, force=bool(force_collectives)) if up is not None else 0
        # 5. this rank's shard, resident
        self.total_units = int(total_units)
        self.start, self.count = shard_frames(self.total_units, self.world, self.rank)
        self.source = source if source is not None else SyntheticSource("gradient")
        self.elapsed_local = None
        self.steps_run = 0
        if not resident:  # the shard goes through in windows (run_windows): nothing of it is loaded here
            self.frames = self.mid = self.up_real = self.up_mid = None
            return
        self.frames = self.source(self.start, self.count + 1, self.width, self.height, self.device)
        want = (self.count + 1, self.height, self.width, 4)
        if tuple(self.frames.shape) != want or self.frames.dtype != torch.uint8 or self.frames.device.type != self.device.type \
                or not self.frames.is_contiguous():
            raise ValueError(f"the frame source must return a contiguous uint8 tensor of shape {want} on {self.device}, "
                             f"got {tuple(self.frames.shape)} {self.frames.dtype} on {self.frames.device}")
        self.mid, self.up_real, self.up_mid = self.pipeline.alloc(self.count, self.device)
        self.elapsed_local = None
        self.steps_run = 0

    # -- collectives (control side only: a barrier and one all_gather of a few numbers)
    def barrier(self) -> None:
        if self._collectives:
            if self.backend == "nccl":
                self._dist.barrier(device_ids=[self.device_index])
            else:
                self._dist.barrier()

    def gather(self, row: dict):
        return gather_rows(row, self.world, self._dist, self._torch, self.comm_device, force=self._collectives)

    def _sync(self) -> None:
        if self.device.type == "cuda":
            self._torch.cuda.synchronize()

    def stream_handle(self) -> int:
        return self._torch.cuda.current_stream().cuda_stream if self.device.type == "cuda" else 0

    def refill(self, source) -> None:
        """The same shard from another source (another pattern of the synthetic stream)."""
        self.source = source
        self.frames.copy_(source(self.start, self.count + 1, self.width, self.height, self.device))

    def step(self) -> None:
        """Enqueue one pass over this rank's shard."""
        p, s = self.pipeline, self.stream_handle()
        if self.count == 0:
            return
        if self.schedule == "unit":
            p.step_unit(self.frames, self.mid, self.up_real, self.up_mid, s)
        elif self.schedule == "fused":
            p.step_fused(self.frames, self.up_real, self.up_mid, s)
        else:
            p.step(self.frames, self.mid, self.up_real, self.up_mid, s)

    def run(self, steps: int = 1, warmup: int = 0, step=None, before_timed=None) -> float:
        """`warmup` untimed passes, then `steps` passes between barrier + device synchronisation on both sides.  Returns this
        rank's seconds; the job's time is the maximum over the ranks (`gather`).  `step`: another pass than self.step (a
        measurement variant); `before_timed`: called after the warm-up has drained, before the first barrier."""
        import time

        step = self.step if step is None else step
        for _ in range(int(warmup)):
            step()
        self._sync()
        if before_timed is not None:
            before_timed()
        self.barrier()
        self._sync()
        t0 = time.perf_counter()
        for _ in range(int(steps)):
            step()
        self._sync()
        self.barrier()
        self.elapsed_local = time.perf_counter() - t0
        self.steps_run = int(steps)
        return self.elapsed_local

    def run_windows(self, window: int, consume=None) -> float:
        """The shard as a STREAM: a frame queue has no end and need not fit in HBM.  The rank's units go through in windows of
        `window` units (+ the overlap frame): window i + 1 is fetched from the source into the second of two frame buffers on a
        side stream while window i is computed, `consume(stream, first_unit, n, mid, up_real, up_mid)` sees each window's outputs
        before the next window overwrites them.  One pass over the shard between two barriers; returns this rank's seconds (source
        and consumer included: that is what a stream costs).  Construct with resident=False.  Same kernels, same bytes per unit as
        the resident step: a unit's digest does not depend on the window (tests/test_sharded_stream.py)."""
        import time

        torch = self._torch
        W = int(window)
        if W < 1:
            raise ValueError("window must be at least one unit")
        gpu = self.device.type == "cuda"
        bufs = [torch.empty((W + 1, self.height, self.width, 4), dtype=torch.uint8, device=self.device) for _ in range(2)]
        mid, up_real, up_mid = self.pipeline.alloc(W, self.device)
        nwin = (self.count + W - 1) // W
        main = torch.cuda.current_stream() if gpu else None
        side = torch.cuda.Stream() if gpu else None
        filled = [torch.cuda.Event() for _ in range(2)] if gpu else None
        computed = [torch.cuda.Event() for _ in range(2)] if gpu else None

        def fetch(wi):  # window wi into buffer wi % 2 (on the side stream: beside the previous window's kernels)
            u0 = wi * W
            m = min(W, self.count - u0)
            if gpu:
                if wi >= 2:
                    side.wait_event(computed[wi % 2])  # its last reader
                with torch.cuda.stream(side):
                    bufs[wi % 2][:m + 1].copy_(self.source(self.start + u0, m + 1, self.width, self.height, self.device))
                filled[wi % 2].record(side)
            else:
                bufs[wi % 2][:m + 1].copy_(self.source(self.start + u0, m + 1, self.width, self.height, self.device))

        p = self.pipeline
        self._sync()
        self.barrier()
        t0 = time.perf_counter()
        if nwin:
            fetch(0)
        for wi in range(nwin):
            u0 = wi * W
            m = min(W, self.count - u0)
            if gpu:
                main.wait_event(filled[wi % 2])
            if wi + 1 < nwin:
                fetch(wi + 1)
            fr = bufs[wi % 2][:m + 1]
            s_ = self.stream_handle()
            if self.schedule == "unit":
                p.step_unit(fr, mid[:m], up_real[:m], up_mid[:m], s_)
            elif self.schedule == "fused":
                p.step_fused(fr, up_real[:m], up_mid[:m], s_)
            else:
                p.step(fr, mid[:m], up_real[:m], up_mid[:m], s_)
            if gpu:
                computed[wi % 2].record(main)
            if consume is not None:
                if gpu:
                    main.synchronize()  # (the consumer reads the outputs; the next window's fetch stays in flight beside it)
                consume(self, self.start + u0, m, mid[:m], up_real[:m], up_mid[:m])
        self._sync()
        self.barrier()
        self.elapsed_local = time.perf_counter() - t0
        self.steps_run = 1
        return self.elapsed_local

    def run_host_fed(self, seconds: float, frames_per_call: int = 12, algorithm: str = "lanczos3") -> float:
        """Mode (ii) of SURVEY.md 8(d)/(e) on this rank, all ranks at the same time: a HOST-resident piece of the stream
        (`frames_per_call` pageable 1080p-shaped frames of this rank's shard) goes through the trait-shaped host entry point
        nus_upscaler_upscale_batch -- one submitting host thread, a retiring thread, the copy pool, three slot streams per GPU,
        H2D / kernel / D2H pipelined, outputs into caller-owned pageable buffers -- for `seconds` of wall time between two
        barriers.  Returns this rank's output frames per second (gather them for the job's rate)."""
        import time

        from . import synthetic as syn

        nb = int(frames_per_call)
        frames = [syn.gradient_frame(self.width, self.height, self.start + k).tobytes() for k in range(nb)]
        u = PyWgpuUpscaler("quality", algorithm, device=self.device_index)
        u.initialize(self.width, self.height, 2 * self.width, 2 * self.height)
        outs = [bytearray(u.output_size) for _ in range(nb)]
        u.upscale_batch_into(frames, outs)  # allocates the slots, touches the buffers
        self.barrier()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            u.upscale_batch_into(frames, outs)
            n += nb
        dt = time.perf_counter() - t0
        self.barrier()
        return n / dt

    def summarize(self, rows) -> dict:
        """What the job did, from the gathered rows (each holds at least elapsed_s): whole-job units and pixels per second
        over the slowest rank's time."""
        elapsed = max(r["elapsed_s"] for r in rows)
        units = self.total_units * self.steps_run
        return {"n_gpus": self.world, "steps": self.steps_run, "units_per_step": self.total_units, "elapsed_s": elapsed,
                "ms_per_step": elapsed / max(1, self.steps_run) * 1e3, "units_per_s": units / elapsed,
                "mpix_per_s": units * self.pipeline.unit_pixels / elapsed / 1e6, "elapsed_by_rank": spread(rows, "elapsed_s", 6),
                "first_unit_by_rank": [int(r["first_unit"]) for r in rows], "units_by_rank": [int(r["units"]) for r in rows],
                "lut_bytes": self.lut_bytes, "backend": self.backend if self.world > 1 else None, "schedule": self.schedule}

    def unit_digests(self, bufs=None, n=None):
        """One number per unit of this rank's shard (or of the `n` units in `bufs` = (mid, up_real, up_mid): a window): the byte sums
        of its three outputs folded together.  A property the domain offers at any size: unit k's digest does not depend on how the
        stream was sharded or windowed (tests/test_sharded_stream.py)."""
        t = self._torch
        M = 4503599627370449  # below 2^52: a digest travels exactly in the float64 rows of `gather`
        out = []
        weights = {}
        mid, up_real, up_mid = bufs if bufs is not None else (self.mid, self.up_real, self.up_mid)
        if self.schedule == "fused":
            mid = None  # the fused schedule never writes the in-between frames (they exist only inside the kernel): nothing to digest
        for k in range(self.count if n is None else n):
            total = 0
            for buf, mult in ((mid, 1), (up_real, 1000003), (up_mid, 998244353)):
                if buf is None:
                    continue
                v = buf[k].reshape(-1).to(t.int64)  # one frame at a time: the int64 view of a 4K frame is 265 MB
                n = v.shape[0]
                if n not in weights:  # position-weighted, so that moved bytes change it: byte * (1 + index mod 251)
                    weights[n] = (t.arange(n, device=v.device, dtype=t.int64) % 251) + 1
                total = (total + int((v * weights[n]).sum().item()) % M * mult) % M
            out.append(total)
        return out

    def close(self) -> None:
        if self._own_group and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._own_group = False


def run_sharded(total_units: int, width: int, height: int, *, steps: int = 1, warmup: int = 0, source=None, sink=None,
                **kwargs) -> dict:
    """The whole job of one rank: build the ShardedStream (placement -> process group -> pipeline -> LUT broadcast -> shard),
    run it, hand this rank's buffers to `sink(stream) -> dict of numbers | None` (save them, check them: outside the timed
    region, before anything is gathered -- nobody leaves early), gather, tear down.  Every rank returns the same summary
    (ShardedStream.summarize) plus its own placement report and the gathered per-rank rows."""
    s = ShardedStream(total_units, width, height, source=source, **kwargs)
    try:
        s.run(steps, warmup)
        row = {"elapsed_s": s.elapsed_local, "first_unit": float(s.start), "units": float(s.count),
               "numa_node": s.placement.get("numa_node"), "bound": 1.0 if s.placement.get("bound") else 0.0}
        # a sink that raises on ONE rank must not leave the others waiting in the gather: its failure travels in the row, every rank
        # learns of it, and the exception is raised again (on the rank it happened on; RuntimeError on the others) after the gather
        failure = None
        row["sink_failed"] = 0.0
        if sink is not None:
            try:
                extra = sink(s)
                if extra:
                    row.update({f"sink_{k}": v for k, v in extra.items()})
            except Exception as e:  # noqa: BLE001 -- re-raised below
                failure = e
                row["sink_failed"] = 1.0
        # (the rows of all ranks must have the same keys: a failed sink's numbers are missing, so only the fixed keys travel then)
        any_failed = max(r["sink_failed"] for r in s.gather({"sink_failed": row["sink_failed"]})) > 0.0
        if any_failed:
            row = {k: v for k, v in row.items() if not k.startswith("sink_") or k == "sink_failed"}
        rows = s.gather(row)
        if failure is not None:
            raise failure
        if any_failed:
            bad = [i for i, r in enumerate(rows) if r["sink_failed"]]
            raise RuntimeError(f"run_sharded: the sink failed on rank(s) {bad}")
        out = s.summarize(rows)
        out.update(rank=s.rank, rows=rows, placement={k: v for k, v in s.placement.items() if not k.startswith("_")})
        return out
    finally:
        s.close()
