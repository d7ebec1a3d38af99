"""bench.py --gpus N started directly: the parent launches N ranks as a child process tree
(torch.distributed.run), relays rank 0's JSON line and the exit code, and stays off the GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

STUB = os.path.join(ROOT, "tests", "helpers", "stub_rank.py")


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_launch_ranks_world2_gloo_relays_rank0_json(nsc):
    bench = _bench()
    rc, lines = bench.launch_ranks(2, STUB, ["--steps", "3"], timeout=300)
    assert rc == 0
    out = [json.loads(ln) for ln in lines if ln.startswith("{")]
    assert len(out) == 1, lines
    assert out[0]["n_gpus"] == 2 and out[0]["max"] == 2.0
    assert out[0]["argv"] == ["--steps", "3"]
    assert out[0]["shard"] == [0, 10]
    assert out[0]["lut"] == len(nsc.build_tables_blob(64, 36, 128, 72))


def test_launch_ranks_propagates_failure():
    bench = _bench()
    rc, _ = bench.launch_ranks(2, STUB, ["--fail"], timeout=300)
    assert rc != 0


def test_parent_process_never_imports_torch_or_the_library():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent must reach the launcher before anything
    that could initialise HIP.  Run it with the launcher stubbed out and look at what got imported."""
    code = (
        "import sys, os; sys.path.insert(0, %r); os.environ.pop('WORLD_SIZE', None)\n"
        "import bench\n"
        "seen = {}\n"
        "def fake(n, script, argv, timeout=None):\n"
        "    seen.update(n=n, script=script, argv=list(argv), torch='torch' in sys.modules,\n"
        "                lib='nu_scaler_amd' in sys.modules)\n"
        "    return 0, ['{\"n_gpus\": %%d}' %% n]\n"
        "bench.launch_ranks = fake\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    assert e.code == 0, e.code\n"
        "assert seen['n'] == 2 and seen['argv'] == ['--gpus', '2', '--steps', '1'], seen\n"
        "assert os.path.basename(seen['script']) == 'bench.py'\n"
        "assert not seen['torch'] and not seen['lib'], seen\n"
        "print('ok')\n" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert '{"n_gpus": 2}' in res.stdout and "ok" in res.stdout


def test_worker_rejects_world_size_mismatch():
    """A rank whose WORLD_SIZE disagrees with --gpus stops before it touches a device."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "--gpus 4 but WORLD_SIZE is 2" in res.stderr
