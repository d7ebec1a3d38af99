"""nus_download / nus_upload (include/nuscaler_hip.h, nu_scaler_amd/transfer.py): the product's road between HBM and a caller's
host buffer -- the end of the reference's upscale() (map, wait, to_vec: nu_scaler_core/src/upscale/mod.rs:1041-1057) for callers
of the *_device entry points.  The pageable pointer never reaches the HIP runtime (docs/d2h_fault_analysis.md)."""
import ctypes

import numpy as np
import pytest

from nu_scaler_amd import _capi, transfer

pytestmark = pytest.mark.gpu

CHUNK = 8 << 20  # nus_transfer.hpp: kTransferChunkBytes; the ring holds 4


def _heap_range():
    """[start of the brk heap, the program break): the heap may be several VMAs in /proc/self/maps (every one labelled [heap])."""
    lo = hi = 0
    with open("/proc/self/maps") as f:
        for ln in f:
            if ln.rstrip().endswith("[heap]"):
                a, b = (int(x, 16) for x in ln.split()[0].split("-"))
                lo = a if lo == 0 else min(lo, a)
                hi = max(hi, b)
    libc = ctypes.CDLL(None)
    libc.sbrk.restype = ctypes.c_void_p
    libc.sbrk.argtypes = [ctypes.c_ssize_t]
    return lo, max(hi, libc.sbrk(0) or 0)


class _Malloc:
    """A block straight from glibc's malloc, with guard bytes on both sides of the part handed out."""

    GUARD = 4096

    def __init__(self, nbytes):
        self.libc = ctypes.CDLL(None)
        self.libc.malloc.restype = ctypes.c_void_p
        self.libc.malloc.argtypes = [ctypes.c_size_t]
        self.libc.free.argtypes = [ctypes.c_void_p]
        self.n = nbytes
        self.base = self.libc.malloc(nbytes + 2 * self.GUARD)
        assert self.base
        self.whole = np.ctypeslib.as_array((ctypes.c_ubyte * (nbytes + 2 * self.GUARD)).from_address(self.base))
        self.whole[:] = 0xC3
        self.addr = self.base + self.GUARD
        self.view = self.whole[self.GUARD:self.GUARD + nbytes]

    def guards_intact(self):
        return bool((self.whole[:self.GUARD] == 0xC3).all() and (self.whole[self.GUARD + self.n:] == 0xC3).all())

    def free(self):
        self.view = self.whole = None
        self.libc.free(self.base)


def test_download_4k_frame_into_a_block_of_the_program_breaks_heap(nsc, oracle_mod):
    """a 1080p -> 4K Lanczos frame computed by upscale_device comes down through nus_download into a 33 MB
    block that glibc carved out of the brk heap -- the kind of destination the runtime's own pageable copy faulted on in round 5
    -- and is compared with the oracle; guard bytes around the destination stay untouched."""
    import torch

    w, h = 1920, 1080
    n = 2 * w * 2 * h * 4
    # a 33 177 600-byte block is under glibc's 32 MiB cap of the dynamic mmap threshold: once one such block has been freed the
    # threshold has risen past it and the next one comes from the program break's heap
    first = _Malloc(n)
    first.free()
    blk = _Malloc(n)
    lo, hi = _heap_range()
    in_heap = lo <= blk.addr and blk.addr + n <= hi
    src = oracle_mod.gen_noise(w, h, 4242)
    st = torch.cuda.current_stream().cuda_stream
    d_in = transfer.to_device(src)
    d_out = torch.empty((2 * h, 2 * w, 4), dtype=torch.uint8, device="cuda:0")
    u = nsc.PyWgpuUpscaler("quality", "lanczos3")
    u.initialize(w, h, 2 * w, 2 * h)
    u.upscale_device(d_in.data_ptr(), d_out.data_ptr(), 1, st)
    # no synchronisation here: the download is ordered after the kernel on the same stream
    assert _capi.lib().nus_download(blk.addr, d_out.data_ptr(), n, st) == _capi.OK, _capi.last_error()
    got = blk.view.reshape(2 * h, 2 * w, 4)
    d = np.abs(got.astype(np.int16) - oracle_mod.lanczos3(src, 2 * w, 2 * h, threads=0).astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    assert blk.guards_intact()
    # the same bytes through a pinned destination (direct DMA, no ring) and through the numpy helper
    pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    assert _capi.lib().nus_download(pinned.data_ptr(), d_out.data_ptr(), n, st) == _capi.OK, _capi.last_error()
    assert np.array_equal(pinned.numpy(), blk.view)
    assert np.array_equal(transfer.to_numpy(d_out), got)
    blk.free()
    if not in_heap:  # (the bytes were checked either way; the point of THIS test is the kind of destination)
        pytest.skip("glibc did not serve the 33 MB block from the brk heap in this process (the break could not grow, or a fixed mmap threshold)")


@pytest.mark.parametrize("nbytes", [1, 4095, CHUNK - 1, CHUNK, CHUNK + 1, 4 * CHUNK, 4 * CHUNK + 12345, 9 * CHUNK + 7])
def test_round_trip_sizes_around_the_chunk_and_ring_boundaries(nsc, nbytes):
    """upload then download of pseudo-random bytes: one byte, just under / at / over one chunk, exactly the ring (4 chunks),
    past it (the ring wraps: chunk k is re-used while k+1.. are still on the wire), odd tails; odd host alignment; guard bytes
    around the host destination and around the device range stay untouched."""
    import torch

    rng = np.random.default_rng(nbytes)
    src = rng.integers(0, 256, nbytes + 3, dtype=np.uint8)[3:]  # (odd alignment of the host source)
    guard = 1 << 16
    d = torch.full((nbytes + 2 * guard,), 0x5A, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    transfer.upload(d.data_ptr() + guard, src, st)
    blk = _Malloc(nbytes + 1)
    assert _capi.lib().nus_download(blk.addr + 1, d.data_ptr() + guard, nbytes, st) == _capi.OK, _capi.last_error()
    assert np.array_equal(blk.view[1:], src) and blk.guards_intact() and blk.view[0] == 0xC3
    whole = transfer.to_numpy(d)
    assert (whole[:guard] == 0x5A).all() and (whole[guard + nbytes:] == 0x5A).all() and np.array_equal(whole[guard:guard + nbytes], src)
    blk.free()


def test_upload_is_stream_ordered_and_the_source_is_reusable_on_return(nsc):
    """nus_upload returns once the source has been staged: overwriting the source immediately afterwards must not change what
    arrives; a kernel enqueued on the same stream afterwards sees the uploaded bytes (no host synchronisation in between)."""
    import torch

    n = 5 * CHUNK + 321
    src = np.arange(n, dtype=np.uint32).view(np.uint8)[:n].copy()
    want = src.copy()
    d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        transfer.upload(d.data_ptr(), src, s.cuda_stream)
        src[:] = 0  # the contract: host_src may be re-used on return
        doubled = d.to(torch.int16) * 2  # enqueued behind the last chunk's DMA on the same stream
        got = transfer.to_numpy(doubled)  # (to_numpy downloads on the current stream: s)
    assert np.array_equal(got, want.astype(np.int16) * 2)


def test_transfer_refuses_what_is_not_device_memory(nsc):
    import torch

    host = np.zeros(64, np.uint8)
    other = np.zeros(64, np.uint8)
    L = _capi.lib()
    assert L.nus_download(host.ctypes.data, other.ctypes.data, 64, None) == _capi.ERR_INVALID_ARGUMENT
    assert "device" in _capi.last_error()
    pinned = torch.empty(64, dtype=torch.uint8, pin_memory=True)
    assert L.nus_upload(pinned.data_ptr(), host.ctypes.data, 64, None) == _capi.ERR_INVALID_ARGUMENT  # pinned HOST memory is not a device destination
    d = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    assert L.nus_download(host.ctypes.data, d.data_ptr(), 0, None) == _capi.OK
    d2 = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    assert L.nus_download(d2.data_ptr(), d.data_ptr(), 64, None) == _capi.ERR_INVALID_ARGUMENT  # a device pointer as the HOST side
    assert "host pointer points into device memory" in _capi.last_error()
    assert L.nus_upload(d.data_ptr(), d2.data_ptr(), 64, None) == _capi.ERR_INVALID_ARGUMENT


def test_host_ranges_record_pins_and_the_librarys_own_pinned_memory(nsc):
    """nus_host_ranges: nus_host_pin adds a live entry and nus_host_unpin removes it (history keeps both events); the transfer
    ring's chunks are live entries of kind 2 once a transfer has run; unpinning twice is refused without asking the runtime."""
    import torch

    class R(ctypes.Structure):
        _fields_ = [("seq", ctypes.c_uint64), ("lo", ctypes.c_size_t), ("hi", ctypes.c_size_t), ("kind", ctypes.c_uint32), ("op", ctypes.c_uint32)]

    L = _capi.lib()

    def snapshot(history):
        buf = (R * 256)()
        k = L.nus_host_ranges(buf, 256, history)
        return [(r.lo, r.hi, r.kind, r.op) for r in buf[:k]]

    d = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
    transfer.to_numpy(d)  # makes sure the ring exists
    ring = [r for r in snapshot(0) if r[2] == 2 and r[1] - r[0] == CHUNK]
    assert len(ring) >= 4
    buf = bytearray(1 << 20)
    with nsc.PinnedBuffer(buf) as b:
        addr = ctypes.addressof((ctypes.c_ubyte * len(b)).from_buffer(b))
        assert (addr, addr + len(b), 1, 1) in snapshot(0)
    assert all(r[0] != addr or r[2] != 1 for r in snapshot(0))
    hist = [r for r in snapshot(1) if r[0] == addr and r[2] == 1]
    assert [r[3] for r in hist[-2:]] == [1, 0]
    assert L.nus_host_unpin(addr) == _capi.ERR_INVALID_ARGUMENT


def test_guard_bands_catch_a_store_past_the_end(nsc):
    """tests/conftest.py `guarded`: a kernel that writes 1 KiB past the end of a device output (the library's own write-only
    probe kernel, told a size 1 KiB larger than the tensor -- the overrun stays inside the guarded allocation) is caught by the
    check every GPU test runs at its end; an exact-size write is not."""
    import torch

    from conftest import guarded

    n = 1 << 20
    st = torch.cuda.current_stream().cuda_stream
    L = _capi.lib()
    src = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    ok = guarded.empty(n, dtype=torch.uint8, device="cuda:0")
    assert L.nus_probe_device(2, src.data_ptr(), ok.data_ptr(), n, 0, st) == _capi.OK, _capi.last_error()
    guarded.assert_intact()
    over = guarded.empty(n, dtype=torch.uint8, device="cuda:0")
    assert L.nus_probe_device(2, src.data_ptr(), over.data_ptr(), n + 1024, 0, st) == _capi.OK, _capi.last_error()
    with pytest.raises(AssertionError, match="guard band of device output #0 .* 1024 bytes behind"):
        guarded.assert_intact()
    under = guarded.empty(n, dtype=torch.uint8, device="cuda:0")
    assert L.nus_probe_device(2, src.data_ptr(), under.data_ptr() - 512, n, 0, st) == _capi.OK, _capi.last_error()
    with pytest.raises(AssertionError, match="512 bytes in front"):
        guarded.assert_intact()


def test_transfers_from_several_threads_take_turns(nsc):
    """Concurrent nus_download / nus_upload calls on one device share the ring behind a mutex: four threads, each round-tripping
    its own 20 MiB pattern eight times on its own stream, nobody sees anybody else's bytes."""
    import threading

    import torch

    n = 20 << 20
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            rng = np.random.default_rng(100 + k)
            d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
            for it in range(8):
                src = rng.integers(0, 256, n, dtype=np.uint8)
                transfer.upload(d.data_ptr(), src, s.cuda_stream)
                back = transfer.download(d.data_ptr(), n, np.empty(n, np.uint8), s.cuda_stream)
                if not np.array_equal(back, src):
                    errors.append((k, it))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []


_SEQUENCE = r"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, %r)
import nu_scaler_amd as nsc
from nu_scaler_amd import _capi
nsc.install_fatal_trace(2)
libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p; libc.malloc.argtypes = [ctypes.c_size_t]
libc.free.argtypes = [ctypes.c_void_p]; libc.sbrk.restype = ctypes.c_void_p; libc.sbrk.argtypes = [ctypes.c_ssize_t]
n = 3840 * 2160 * 4
L = _capi.lib()
st = torch.cuda.current_stream().cuda_stream
first = libc.malloc(n); libc.free(first)          # raises glibc's dynamic mmap threshold past one 4K frame
seen = []
for step in range(3):
    d = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda:0")
    want = d.sum().item()
    p = libc.malloc(n)                            # a block of the program break's heap
    brk_before = libc.sbrk(0)
    assert L.nus_download(p, d.data_ptr(), n, st) == _capi.OK, _capi.last_error()
    got = int(np.ctypeslib.as_array((ctypes.c_ubyte * n).from_address(p)).sum(dtype=np.int64))
    assert got == want, (step, got, want)
    libc.free(p)
    libc.malloc_trim(0)                           # the block's pages go back to the kernel (the break lowered, or MADV_DONTNEED inside the heap)
    lo = (p + 4095) & ~4095
    pages = (p + n - lo) // 4096
    vec = (ctypes.c_ubyte * pages)()
    assert libc.mincore(ctypes.c_void_p(lo), ctypes.c_size_t(pages * 4096), vec) in (0, -1)
    resident = sum(v & 1 for v in vec) if libc.sbrk(0) >= lo + pages * 4096 else 0   # (beyond the lowered break: unmapped = gone)
    seen.append((p, brk_before, libc.sbrk(0), resident, pages))
same_address = seen[0][0] == seen[1][0] == seen[2][0]
pages_gone = all(res * 10 < pages for _, _, _, res, pages in seen)
print("SEQUENCE ok same_address=%%s pages_gone=%%s resident=%%s" %% (same_address, pages_gone, [r[3] for r in seen]))
"""


def test_download_into_a_heap_block_that_was_freed_trimmed_and_regrown(nsc):
    """the hypothesised sequence of the round-5 fault, deterministically, in a child process -- but through the PRODUCT's
    road: a 33 MB block of the brk heap as the destination of a device-to-host transfer, free, malloc_trim (the pages go back to the
    kernel), the heap regrown, a block at the same address as the destination again, three times over.  With nus_download no
    registration of the caller's pages exists that could go stale: every round's bytes are right.  (The runtime's own pageable copy is
    deliberately NOT put through this: docs/d2h_fault_analysis.md.)"""
    import subprocess
    import sys

    from conftest import ROOT

    run = subprocess.run([sys.executable, "-c", _SEQUENCE % ROOT], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "SEQUENCE ok" in run.stdout, run.stdout[-1500:] + run.stderr[-3000:]
    # what makes it the sequence in question: the block's pages really went back to the kernel in between (and, normally, the same address came back)
    assert "pages_gone=True" in run.stdout, run.stdout
