"""Device tensor -> host through PINNED staging, for the harness processes (tests, bench.py, smoke()).

Why: twice (round 4, round 5) a long-lived test process died inside `Tensor.cpu()` of a 4K frame right after the CPU oracle had run --
round 5's log names it: ROCr `Memory access fault by GPU node-2 ... on address 0x5b7c... Reason: Write access to a read-only page`, a
HOST address in the process's brk heap, i.e. the runtime's own device-to-host copy into PAGEABLE memory faulted.  For pageable
destinations above ~1 MiB the HIP runtime pins the caller's pages on the fly and keeps such registrations for a while; a frame-sized
block that glibc carved from the program break's heap, freed, trimmed away and grew back at the same address meets a registration
whose pages are gone.  The library's own host path never takes that road (it stages through memory it allocated with hipHostMalloc
and copies with CPU threads, nus_host.cpp), and neither do the harness processes now: device -> pinned tensor (torch's caching host
allocator, hipHostMalloc underneath) -> numpy copy.  profiles/r05_gpu_fault_during_pageable_d2h.txt, docs/abort_r04_analysis.md.
"""
from __future__ import annotations

_MIN_BYTES = 1 << 18


def to_host(t):
    """A CPU tensor with the contents of `t` (CUDA tensors of >= 256 KiB go through a pinned staging tensor)."""
    import torch

    if not t.is_cuda:
        return t
    if t.numel() * t.element_size() < _MIN_BYTES:
        return torch.Tensor.cpu(t) if not hasattr(torch.Tensor, "_nus_plain_cpu") else torch.Tensor._nus_plain_cpu(t)
    stage = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    stage.copy_(t.contiguous())  # synchronous for the caller: copy_ without non_blocking waits for the stream
    return stage.clone()         # pageable copy made by the CPU; the pinned block goes back to torch's cache


def to_numpy(t):
    return to_host(t).numpy()


def route_tensor_cpu_through_pinned_staging() -> None:
    """Harness switch: `Tensor.cpu()` of a large CUDA tensor takes `to_host` from now on (idempotent)."""
    import torch

    if hasattr(torch.Tensor, "_nus_plain_cpu"):
        return
    plain = torch.Tensor.cpu
    torch.Tensor._nus_plain_cpu = plain

    def cpu(self, *args, **kwargs):
        if args or kwargs or not self.is_cuda or self.numel() * self.element_size() < _MIN_BYTES:
            return plain(self, *args, **kwargs)
        return to_host(self)

    torch.Tensor.cpu = cpu
