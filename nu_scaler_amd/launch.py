"""Start the ranks of a one-node job: one process per GPU.

`launch_ranks` runs `python -m torch.distributed.run --nproc-per-node N <script | -m module> args` as a CHILD process and waits
for it -- never an exec: a process that has initialised HIP must not be replaced, and the launching process itself stays off the
GPU (it imports neither torch.cuda nor libnuscaler_hip.so), so the ranks are its grandchildren.  Rendezvous on 127.0.0.1 (a
container's hostname may not resolve).  The reference has no counterpart: it drives one adapter from one process
(nu_scaler_core/src/gpu/detector.rs:136-165).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple

LAUNCHER_OMP_MARK = "NUS_OMP_THREADS_FROM_LAUNCHER"  # placement.bind_rank: this OMP_NUM_THREADS is the launcher's default, not the operator's


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environment(environ=None) -> dict:
    """The environment the ranks get: dmabuf IPC for RCCL on this driver, and an OpenMP default that each rank re-sizes from its
    CPU share (torchrun's own default is 1) unless the operator set one."""
    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "OMP_NUM_THREADS" not in env:
        env["OMP_NUM_THREADS"] = "4"
        env[LAUNCHER_OMP_MARK] = "1"
    return env


def launch_ranks(nproc: int, script: str, script_args: Sequence[str], timeout: Optional[float] = None, module: bool = False,
                 environ=None) -> Tuple[int, List[str]]:
    """Start `nproc` ranks of `script` (a path, or a module name with module=True) and wait.  Returns (returncode, stdout
    lines); the ranks' stderr goes to this process's stderr."""
    env = rank_environment(environ)
    if module:  # the package must be importable from the ranks whatever their working directory
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += (["-m", script] if module else [script]) + list(script_args)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout)
    return res.returncode, res.stdout.splitlines()


def under_launcher(environ=None) -> bool:
    """True inside a rank torchrun started (RANK and WORLD_SIZE are set)."""
    env = os.environ if environ is None else environ
    return "RANK" in env and "WORLD_SIZE" in env
