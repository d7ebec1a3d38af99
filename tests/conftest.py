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
