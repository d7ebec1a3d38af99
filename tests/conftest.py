import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: load it before libnuscaler_hip.so pulls in /opt/rocm's,
    import torch  # noqa: F401  (otherwise torch.cuda later reports "No HIP GPUs are available")
except Exception:  # pragma: no cover
    torch = None

# Round 4: one full GPU run died of SIGABRT with nothing but "Fatal Python error: Aborted" in its log.  Why nothing: pytest's
# fd-level capture holds file descriptor 2 while a test runs, so whatever the aborting runtime said (glibc, ROCr, libstdc++ all
# write to fd 2) went into a capture file that died with the process.  pytest.ini now runs the suite with --capture=sys, and the
# handler installed below (the library's nus_install_fatal_trace) writes the NATIVE backtrace of the aborting thread -- which
# library called abort --, the host ranges the library holds and /proc/self/maps in front of faulthandler's Python frames.  (LIBC_FATAL_STDERR_ only matters to glibc < 2.27; kept for those.)
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
import faulthandler  # noqa: E402

faulthandler.enable(file=sys.__stderr__, all_threads=True)  # (sys.stderr is pytest's capture object under --capture=sys)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # after every plugin's configure step (pytest's own faulthandler plugin included): outermost handler, real stderr
    import nu_scaler_amd

    if os.path.exists(nu_scaler_amd._capi.LIB_PATH):
        nu_scaler_amd.install_fatal_trace(2)
    # Device tensors come down through nu_scaler_amd.transfer.to_numpy (`fetch` in the test modules: nus_download, the product's
    # own road), never Tensor.cpu(): the HIP runtime's copy into PAGEABLE memory is what died twice in long test sessions
    # (ROCr: "Write access to a read-only page" at a brk-heap address; docs/d2h_fault_analysis.md).  tests/test_no_tensor_cpu.py
    # keeps it that way.


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def nsc():
    """The product package; builds libnuscaler_hip.so with hipcc if it is missing."""
    import nu_scaler_amd

    if not os.path.exists(nu_scaler_amd._capi.LIB_PATH):
        nu_scaler_amd.build()
    nu_scaler_amd._capi.lib()
    return nu_scaler_amd


@pytest.fixture(scope="session")
def golden():
    from _png import read_png

    return {
        "test_input": read_png(os.path.join(GOLDEN, "ref_test_input.png")),
        "test_output": read_png(os.path.join(GOLDEN, "ref_test_output.png")),
        "interp_half": read_png(os.path.join(GOLDEN, "ref_interp_half.png")),
    }


class _Guarded:
    """Device tensors for the OUTPUTS of the `*_device` entry points, carved out of a larger allocation with 64 KiB of a poison
    byte in front of and behind the tensor (a kernel that stored past a caller's buffer once passed its first
    test shape because the overrun landed in memory the process owned).  Same call shapes as torch.empty / zeros / full /
    *_like; `assert_intact()` synchronises and checks every guard handed out since the last check -- the autouse fixture below
    calls it after every test, tests may call it earlier."""

    GUARD = 1 << 16
    POISON = 0xA7

    def __init__(self):
        self._live = []

    def _carve(self, shape, dtype, device, fill):
        import math

        import torch

        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        shape = tuple(int(s) for s in shape)
        dtype = dtype or torch.float32
        n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        pad = (-n) % 256
        flat = torch.full((2 * self.GUARD + n + pad,), self.POISON, dtype=torch.uint8, device=device)
        payload = flat[self.GUARD:self.GUARD + n].view(dtype).view(shape)
        if fill is not None:
            payload.fill_(fill)
        self._live.append((flat, n))
        return payload

    def empty(self, *shape, dtype=None, device=None):
        return self._carve(shape, dtype, device, None)  # (poison inside as well: an output the kernel must write completely)

    def zeros(self, *shape, dtype=None, device=None):
        return self._carve(shape, dtype, device, 0)

    def full(self, shape, value, dtype=None, device=None):
        return self._carve((shape,) if isinstance(shape, int) else tuple(shape), dtype, device, value)

    def empty_like(self, t):
        return self._carve(tuple(t.shape), t.dtype, t.device, None)

    def zeros_like(self, t):
        return self._carve(tuple(t.shape), t.dtype, t.device, 0)

    def like(self, tensors):
        """Zero-filled guarded tensors with the shapes of `tensors` (what FramePipeline.alloc returns)."""
        return tuple(self.zeros_like(t) for t in tensors)

    def assert_intact(self):
        if not self._live:
            return
        import torch

        torch.cuda.synchronize()
        live, self._live = self._live, []
        for i, (flat, n) in enumerate(live):
            front, back = flat[:self.GUARD], flat[self.GUARD + n:]
            bad_f, bad_b = int((front != self.POISON).sum()), int((back != self.POISON).sum())
            assert bad_f == 0 and bad_b == 0, (f"guard band of device output #{i} ({n} bytes) was written: {bad_f} bytes in front, "
                                               f"{bad_b} bytes behind")


guarded = _Guarded()


@pytest.fixture(autouse=True)
def _device_output_guards():
    yield
    guarded.assert_intact()
