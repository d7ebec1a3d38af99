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
    # the per-rank rows every rank holds after the gather: rank order, None preserved, the spread rank 0 prints
    rows = out[0]["rows"]
    assert [r["elapsed_s"] for r in rows] == [1.0, 2.0] and [r["first_frame"] for r in rows] == [0.0, 10.0]
    assert rows[0]["sclk_MHz"] == 2100.0 and rows[1]["sclk_MHz"] is None
    assert out[0]["elapsed"] == {"min": 1.0, "max": 2.0, "by_rank": [1.0, 2.0]}
    assert out[0]["sclk"] == {"min": 2100.0, "max": 2100.0, "by_rank": [2100.0, None]}
    assert all(r["cpus_per_rank"] >= 1 for r in rows)


def test_launch_ranks_world8_gloo_every_by_rank_list_has_eight_entries(nsc):
    """the launcher and the gather with WORLD_SIZE = 8 on the CPU -- stdout is exactly ONE JSON line (what
    the driver parses), every per-rank list in it has eight entries in rank order, the shard arithmetic covers 80 frames without
    gap or overlap, the LUT reached everybody, and the copy pool of a rank with an eighth of this box's CPUs has no workers."""
    bench = _bench()
    rc, lines = bench.launch_ranks(8, STUB, ["--steps", "2"], timeout=600)
    assert rc == 0
    text = [ln for ln in lines if ln.strip()]
    assert len(text) == 1 and text[0].startswith("{"), text[:6]
    out = json.loads(text[0])
    assert out["n_gpus"] == 8 and out["max"] == 8.0 and out["shard"] == [0, 10]
    assert out["lut"] == len(nsc.build_tables_blob(64, 36, 128, 72))
    rows = out["rows"]
    assert len(rows) == 8 and [r["first_frame"] for r in rows] == [10.0 * r for r in range(8)]
    assert [r["elapsed_s"] for r in rows] == [1.0 + r for r in range(8)]
    for key in ("elapsed", "sclk"):
        assert len(out[key]["by_rank"]) == 8, key
    assert out["elapsed"] == {"min": 1.0, "max": 8.0, "by_rank": [1.0 + r for r in range(8)]}
    assert out["sclk"]["by_rank"] == [2100.0] + [None] * 7
    assert all(r["cpus_per_rank"] >= 1 for r in rows)


def test_launch_ranks_propagates_failure():
    bench = _bench()
    rc, _ = bench.launch_ranks(2, STUB, ["--fail"], timeout=300)
    assert rc != 0


def test_parent_process_never_imports_torch_or_the_library():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent must reach the product's launcher (nu_scaler_amd.launch) and
    start its child without anything that could initialise HIP -- torch is not imported, libnuscaler_hip.so is not loaded
    (importing the package loads neither).  Run with the child process faked and look at what got imported."""
    code = (
        "import sys, os; sys.path.insert(0, %r); os.environ.pop('WORLD_SIZE', None); os.environ.pop('RANK', None)\n"
        "import bench\n"
        "from nu_scaler_amd import launch\n"
        "seen = {}\n"
        "class R: returncode = 0; stdout = '{\"n_gpus\": 2}\\n'\n"
        "def fake_run(cmd, **kw):\n"
        "    import nu_scaler_amd\n"
        "    seen.update(cmd=list(cmd), torch='torch' in sys.modules, lib=nu_scaler_amd._capi._lib is not None, env=kw['env'])\n"
        "    return R()\n"
        "launch.subprocess.run = fake_run\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    assert e.code == 0, e.code\n"
        "cmd = seen['cmd']\n"
        "assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=2' in cmd and '127.0.0.1' in cmd, cmd\n"
        "assert os.path.basename(cmd[-5]) == 'bench.py' and cmd[-4:] == ['--gpus', '2', '--steps', '1'], cmd\n"
        "assert not seen['torch'] and not seen['lib'], seen\n"
        "assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and 'OMP_NUM_THREADS' in seen['env']\n"
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


def test_rank_check_regenerates_its_shard_and_never_raises(oracle_mod):
    """bench.check_timed_outputs on host tensors: a rank's frames are regenerated from their position in the GLOBAL stream, so a
    rank holding the wrong shard fails `input_ok`; a damaged output fails `ok`; nothing raises (the summaries are gathered
    first, then every rank exits)."""
    import numpy as np
    import torch

    bench = _bench()
    w, h, first = 64, 24, 300  # rank 1 of a 300-units-per-rank job
    src = [oracle_mod.gen_gradient(w, h, first + k) for k in range(3)]
    frames = torch.from_numpy(np.stack(src))
    mids = [oracle_mod.warp_blend(src[k], src[k + 1], None, 0.5) for k in range(2)]
    mid = torch.from_numpy(np.stack(mids))
    up_real = torch.from_numpy(np.stack([oracle_mod.lanczos3(src[k], 2 * w, 2 * h) for k in range(2)]))
    up_mid = torch.from_numpy(np.stack([oracle_mod.lanczos3(m, 2 * w, 2 * h) for m in mids]))
    report, summ = bench.check_timed_outputs(frames, mid, up_real, up_mid, [0, 1], w, h, first_frame=first, threads=2)
    assert summ == {"ok": 1.0, "input_ok": 1.0, "mid_exact": 1.0, "max_abs_diff": 0.0, "frac_differing": 0.0, "frames": 2.0}
    assert {r["buffer"] for r in report} == {"input", "mid", "up_real", "up_mid"}
    _, wrong_shard = bench.check_timed_outputs(frames, mid, up_real, up_mid, [0, 1], w, h, first_frame=0, threads=2)
    assert wrong_shard["input_ok"] == 0.0 and wrong_shard["ok"] == 0.0
    bad = up_mid.clone()
    bad[1, 5, 7, 2] ^= 0x40
    _, damaged = bench.check_timed_outputs(frames, mid, up_real, bad, [0, 1], w, h, first_frame=first, threads=2)
    assert damaged["ok"] == 0.0 and damaged["input_ok"] == 1.0 and damaged["max_abs_diff"] >= 2
    bad_mid = mid.clone()
    bad_mid[0, 0, 0, 0] ^= 1
    _, damaged = bench.check_timed_outputs(frames, bad_mid, up_real, up_mid, [0], w, h, first_frame=first, threads=2)
    assert damaged["ok"] == 0.0 and damaged["mid_exact"] == 0.0


def test_clock_sampler_parses_rocm_smi_and_sysfs(tmp_path):
    bench = _bench()
    text = ("GPU[3]\t\t: fclk clock level: 0: (2000Mhz)\nGPU[3]\t\t: mclk clock level: 3: (1900Mhz)\n"
            "GPU[3]\t\t: sclk clock level: 1: (2338Mhz)\nGPU[3]\t\t: socclk clock level: 0: (28Mhz)\n"
            "GPU[3]\t\t: Current Socket Graphics Package Power (W): 1350.0\n")
    assert bench.ClockSampler._parse(text) == (2338, 1350.0, 1900, 2000)
    f = tmp_path / "pp_dpm_sclk"
    f.write_text("0: 132Mhz\n1: 2120Mhz *\n2: 2400Mhz\n")
    assert bench.ClockSampler._dpm_current(str(f)) == 2120
    assert bench.ClockSampler._dpm_current(str(tmp_path / "missing")) is None
    assert bench.ClockSampler.mean([(0.0, 2000, None), (0.6, 2100, 1300.0)], 1) == 2050
    assert bench.ClockSampler.mean([(0.0, 2000, None)], 2, 1) is None
