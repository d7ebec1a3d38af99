"""Loads tests/helpers/abort_trace.c (built on demand with gcc) and installs its fatal-signal handler: the native backtrace of
an abort() / SIGSEGV goes to the real stderr before faulthandler dumps the Python frames.  Test infrastructure only."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libaborttrace.so")  # (not abort_trace.so: Python would import that as this module)
_SRC = os.path.join(_HERE, "abort_trace.c")
_keep = None


def install(fd: int = 2) -> bool:
    """True when the handler is in place.  Never raises: a box without gcc just keeps faulthandler's Python-only report."""
    global _keep
    if _keep is not None:
        return True
    try:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
            tmp = f"{_SO}.{os.getpid()}.tmp"
            subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-shared", "-rdynamic", "-o", tmp, _SRC], check=True, capture_output=True)
            os.replace(tmp, _SO)  # atomic: several ranks may get here together
        lib = ctypes.CDLL(_SO)
        lib.abort_trace_install.argtypes = [ctypes.c_int]
        lib.abort_trace_install.restype = ctypes.c_int
        if lib.abort_trace_install(int(fd)) != 0:
            return False
        _keep = lib
        return True
    except Exception:
        return False
