"""bench.py with two ranks (gloo) on the one GPU of the box: the N > 1 path end to end -- placement before the first HIP call,
the LUT broadcast, every rank's check of its own shard, the gathered per-rank numbers in rank 0's line.  (RCCL cannot put two
ranks on one GPU: the collective backend is gloo here; tests/test_rccl_one_rank.py drives RCCL itself.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_rehearsal_reports_every_rank(nsc):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--force-device", "0", "--units", "8",
           "--steps", "3", "--warmup", "1", "--sustained-seconds", "1", "--host-fed-seconds", "0.5"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    c, r = d["config"], d["roofline"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert c["lut_broadcast_bytes"] > 0
    # placement: both ranks report; on a box whose sysfs names the GPU's node they are bound to disjoint CPUs
    p = c["placement"]
    assert len(p["bound_by_rank"]) == 2 and len(p["cpus_per_rank"]) == 2 and all(n >= 1 for n in p["cpus_per_rank"])
    assert p["rank0"]["local_world"] == 2 and (p["rank0"]["bound"] or p["rank0"]["why_not"])
    assert all(v in (True, None) for v in p["gpu_pci_address_confirmed_by_hip_by_rank"])
    # every rank checked frames of ITS shard
    chk = c["per_rank_check"]
    assert chk["all_ok"] and [x["rank"] for x in chk["by_rank"]] == [0, 1]
    assert [x["first_stream_frame"] for x in chk["by_rank"]] == [0, 8]
    assert all(x["ok"] and x["input_ok"] and x["mid_bit_exact"] and x["frames_checked"] == 2 and x["max_abs_diff"] <= 1 for x in chk["by_rank"])
    # every rank's numbers
    assert len(c["ms_per_step_by_rank"]["by_rank"]) == 2 and c["ms_per_step_by_rank"]["max"] == d["ms_per_step"]
    assert len(r["frac_by_rank"]["by_rank"]) == 2 and len(c["per_rank"]["bracket_ms"]["by_rank"]) == 2
    assert len(c["per_rank"]["sustained_ms_per_step"]["by_rank"]) == 2
    assert set(r["copy_ceiling"]["by_rank"]) >= {"stream_copy_float4_GBps", "one_read_four_writes_GBps", "hipMemcpyDtoDAsync_GBps"}
    hf = c["host_fed"]
    assert len(hf["frames_4k_per_s_per_gpu"]["by_rank"]) == 2 and len(hf["fed_from_numa_node_by_rank"]) == 2
    assert "cpu_baseline" not in d and r["traffic"] is None  # N = 1 only legs


def test_one_rank_rehearsal_over_rccl_keeps_stdout_to_one_line(nsc):
    """bench.py --force-collectives: N = 1, but the communicator is created (backend nccl = RCCL, device_id = the GPU) and every
    collective of the N > 1 path is issued on it.  RCCL prints a five-line version banner on STDOUT when rank 0 creates its
    communicator: the product points fd 1 at fd 2 for that moment, so the job's stdout stays what the driver parses -- ONE JSON
    line -- and the banner is in stderr."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-collectives", "--units", "20", "--steps", "3", "--warmup", "1",
           "--sustained-seconds", "0", "--config3-seconds", "0", "--no-pmc", "--no-cpu-baseline", "--no-extras"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out_lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1 and out_lines[0].startswith("{"), out_lines[:8]
    d = json.loads(out_lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["lut_broadcast_bytes"] > 0  # the tables went through the broadcast
    assert "RCCL version" in res.stderr
