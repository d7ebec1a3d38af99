import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: load it before libnuscaler_hip.so pulls in /opt/rocm's,
    import torch  # noqa: F401  (otherwise torch.cuda later reports "No HIP GPUs are available")
except Exception:  # pragma: no cover
    torch = None

# A fatal message of glibc (heap corruption, stack smashing) goes to the controlling terminal unless this is set: one full GPU run
# of round 4 died of SIGABRT with nothing but "Fatal Python error: Aborted" in its log.  With it the cause would have been in the log.
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
import faulthandler  # noqa: E402

faulthandler.enable(all_threads=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
