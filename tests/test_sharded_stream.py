"""The sharded frame-queue stream as a product entry point (nu_scaler_amd.stream.ShardedStream / run_sharded, `python -m
nu_scaler_amd.cli stream`): placement -> process group -> pipeline -> LUT broadcast -> shard -> step loop -> gather.

CPU, gloo, world_size 2: the rank loop on host tensors with an oracle-backed stand-in for the pipeline (the product has no CPU
path: without a HIP device the real pipeline refuses, which is tested too).  GPU: the real pipeline, and the property the domain
offers at any size -- a unit's outputs do not depend on how the stream was sharded."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from nu_scaler_amd.transfer import to_device as put, to_numpy as fetch  # host <-> HBM through nus_upload / nus_download, never
# torch's pageable copies (docs/d2h_fault_analysis.md)

HELPER = os.path.join(ROOT, "tests", "helpers", "sharded_rank.py")


def _launch(world, extra=(), timeout=600):
    sys.path.insert(0, ROOT)
    from nu_scaler_amd import launch

    rc, lines = launch.launch_ranks(world, HELPER, list(extra), timeout=timeout)
    out = [json.loads(ln) for ln in lines if ln.startswith("{")]
    return rc, out


def test_world2_gloo_rank_loop_on_host_tensors(nsc, oracle_mod):
    """Two ranks, 7 units (shards of 4 and 3, each with its overlap frame): every rank checks the three outputs of EVERY unit of
    its shard against the oracle, the tables reach rank 1 through the broadcast, the summary is the same on both ranks, and the
    per-unit digests equal those of the same stream run by one rank."""
    rc, out = _launch(2, ["--units-total", "7", "--steps", "2", "--warmup", "1"])
    assert rc == 0 and len(out) == 1, out
    two = out[0]
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["units_per_step"] == 7
    assert two["first_unit_by_rank"] == [0, 4] and two["units_by_rank"] == [4, 3]
    assert two["checked_units_by_rank"] == [4, 3] and two["mismatches_by_rank"] == [0, 0]
    assert two["lut_bytes"] == len(nsc.build_tables_blob(48, 20, 96, 40)) and two["tables_imported_by_rank"] == [0, 1]
    assert two["summary_equal_on_all_ranks"] is True
    assert two["elapsed_s"] == pytest.approx(max(two["elapsed_by_rank"]["by_rank"]), abs=1e-6) and two["elapsed_s"] > 0
    assert two["mpix_per_s"] == pytest.approx(7 * 2 * two["unit_pixels"] / two["elapsed_s"] / 1e6)
    rc, out = _launch(1, ["--units-total", "7", "--steps", "1"])
    assert rc == 0 and len(out) == 1
    one = out[0]
    assert one["n_gpus"] == 1 and one["units_by_rank"] == [7] and one["lut_bytes"] == 0
    assert one["unit_digests"] == two["unit_digests"] and len(one["unit_digests"]) == 7
    assert len(set(one["unit_digests"])) == 7  # the stream moves: no two units alike


def test_world3_ragged_shards_and_an_empty_rank(nsc):
    """More ranks than units: a rank with an empty shard still takes part in every collective and reports zero units."""
    rc, out = _launch(3, ["--units-total", "2", "--steps", "1"])
    assert rc == 0 and len(out) == 1, out
    assert out[0]["units_by_rank"] == [1, 1, 0] and out[0]["mismatches_by_rank"] == [0, 0, 0]
    assert len(out[0]["unit_digests"]) == 2


def test_the_shard_in_windows_is_the_same_stream(nsc):
    """ShardedStream.run_windows: a stream need not fit in HBM -- each rank's shard goes through in windows (+ the overlap frame),
    the next window fetched while the current one is computed.  Two ranks, 11 units, windows of 4: 6 + 5 units in 2 + 2 windows
    (4 + 2, 4 + 1), every unit of every window against the oracle, and the digests of the 11 units equal those of the resident
    one-rank run: a unit's outputs depend neither on the sharding nor on the window."""
    rc, out = _launch(2, ["--units-total", "11", "--window", "4"])
    assert rc == 0 and len(out) == 1, out
    win = out[0]
    assert win["units_by_rank"] == [6, 5] and win["windows_by_rank"] == [2, 2] and win["mismatches_by_rank"] == [0, 0]
    rc, out = _launch(1, ["--units-total", "11"])
    assert rc == 0 and out[0]["unit_digests"] == win["unit_digests"] and len(win["unit_digests"]) == 11
    rc, out = _launch(1, ["--units-total", "11", "--window", "16"])  # one window larger than the shard
    assert rc == 0 and out[0]["unit_digests"] == win["unit_digests"] and out[0]["windows_by_rank"] == [1]


def test_world8_gloo_ragged(nsc):
    """the rank arithmetic of the 8-GPU job, end to end, before the driver's first 8-GPU contact --
    launch_ranks(8) -> placement -> process group -> LUT broadcast to seven ranks -> shard_frames -> step loop -> gather, on
    host tensors over gloo.  301 units: five shards of 38 and three of 37 (each with its overlap frame), every unit of every
    shard against the oracle on its own rank, the 301 digests equal to the one-rank run's.  5 units: five ranks with one unit,
    THREE EMPTY ranks that still take part in every collective.  A sink that fails on rank 6 ends the job on all eight."""
    import time

    rc, out = _launch(8, ["--units-total", "301", "--steps", "1"], timeout=900)
    assert rc == 0 and len(out) == 1, out
    eight = out[0]
    assert eight["n_gpus"] == 8 and eight["units_per_step"] == 301
    shards = [nsc.shard_frames(301, 8, r) for r in range(8)]
    assert [c for _, c in shards] == [38] * 5 + [37] * 3 and [s for s, _ in shards] == [0, 38, 76, 114, 152, 190, 227, 264]
    assert eight["units_by_rank"] == [c for _, c in shards] and eight["first_unit_by_rank"] == [s for s, _ in shards]
    assert eight["checked_units_by_rank"] == eight["units_by_rank"] and eight["mismatches_by_rank"] == [0] * 8
    assert eight["lut_bytes"] == len(nsc.build_tables_blob(48, 20, 96, 40)) and eight["tables_imported_by_rank"] == [0] + [1] * 7
    assert eight["summary_equal_on_all_ranks"] is True
    assert len(eight["elapsed_by_rank"]["by_rank"]) == 8 and eight["elapsed_s"] == pytest.approx(max(eight["elapsed_by_rank"]["by_rank"]), abs=1e-6)
    rc, out = _launch(1, ["--units-total", "301", "--steps", "1"], timeout=900)
    assert rc == 0 and len(out) == 1
    d = eight["unit_digests"]
    assert out[0]["unit_digests"] == d and len(d) == 301
    # (frame k is the gradient shifted by k mod w: the 48-pixel-wide stream repeats after 48 units, across shard borders too)
    assert len(set(d)) == 48 and all(d[k] == d[k + 48] for k in range(301 - 48))
    rc, out = _launch(8, ["--units-total", "5", "--steps", "2", "--warmup", "1"], timeout=600)
    assert rc == 0 and len(out) == 1, out
    five = out[0]
    assert five["units_by_rank"] == [1] * 5 + [0] * 3 and five["mismatches_by_rank"] == [0] * 8
    assert five["checked_units_by_rank"] == [1] * 5 + [0] * 3 and five["tables_imported_by_rank"] == [0] + [1] * 7
    assert five["unit_digests"] == eight["unit_digests"][:5]
    t0 = time.time()
    rc, out = _launch(8, ["--units-total", "20", "--sink-fails-on", "6"], timeout=300)
    assert rc != 0 and out == [] and time.time() - t0 < 250


def test_a_sink_that_raises_on_one_rank_ends_the_job(nsc):
    """run_sharded: the failure of one rank's sink travels through the gather -- every rank raises (the failing one its own
    exception), nobody is left waiting in a collective, the launcher reports a non-zero exit code within seconds."""
    import time

    t0 = time.time()
    rc, out = _launch(2, ["--units-total", "5", "--sink-fails-on", "1"], timeout=240)
    assert rc != 0 and out == [] and time.time() - t0 < 200


def test_frame_source_shape_is_checked(nsc):
    import torch

    with pytest.raises(ValueError, match="frame source must return"):
        nsc.ShardedStream(3, 32, 8, source=lambda first, n, w, h, dev: torch.zeros((n, h, w, 3), dtype=torch.uint8), device_kind="cpu",
                          backend="gloo", bind=False, pipeline_factory=lambda *a: None)


def test_cli_stream_without_a_gpu_fails_loudly(nsc):
    """No CPU fallback: `stream` on a box without a HIP device says so and exits non-zero -- alone and through the launcher."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "nu_scaler_amd.cli", "stream", "--units", "2", "--width", "64", "--height", "32"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0 and "needs a HIP device" in res.stderr
    res = subprocess.run([sys.executable, "-m", "nu_scaler_amd.cli", "stream", "--gpus", "2", "--backend", "gloo", "--units", "2",
                          "--width", "64", "--height", "32"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode != 0 and "needs a HIP device" in res.stderr and "{" not in res.stdout


def test_synthetic_source_is_shard_invariant(nsc):
    import torch

    S = nsc.stream
    for pattern in ("gradient", "noise"):
        src = S.SyntheticSource(pattern)
        whole = src(5, 40, 32, 8, torch.device("cpu"))
        parts = torch.cat([src(5, 17, 32, 8, torch.device("cpu")), src(22, 23, 32, 8, torch.device("cpu"))])
        assert torch.equal(whole, parts)
    assert np.array_equal(S.SyntheticSource("gradient")(3, 1, 64, 16, torch.device("cpu"))[0].numpy(),
                          nsc.synthetic.gradient_frame(64, 16, 3))


@pytest.mark.gpu
def test_sharded_stream_one_rank_against_the_oracle(nsc, oracle_mod):
    """World of one on the GPU, the real pipeline: every unit's in-between frame bit-exact, both 4K frames within 1 LSB."""
    w, h, n = 496, 270, 5
    s = nsc.ShardedStream(n, w, h, bind=False)
    try:
        s.run(steps=2, warmup=1)
        rows = s.gather({"elapsed_s": s.elapsed_local, "first_unit": float(s.start), "units": float(s.count)})
        summ = s.summarize(rows)
        assert summ["n_gpus"] == 1 and summ["units_by_rank"] == [n] and summ["steps"] == 2
        assert s.placement["gpu_bdf_verified"] in (True, False, None) and "_replan" not in s.placement
        for k in range(n):
            a, b = oracle_mod.gen_gradient(w, h, k), oracle_mod.gen_gradient(w, h, k + 1)
            assert np.array_equal(fetch(s.frames[k]), a)
            m = oracle_mod.warp_blend(a, b, None, 0.5)
            assert np.array_equal(fetch(s.mid[k]), m)
            for got, src in ((s.up_real, a), (s.up_mid, m)):
                d = np.abs(fetch(got[k]).astype(np.int16) - oracle_mod.lanczos3(src, 2 * w, 2 * h, threads=0).astype(np.int16))
                assert d.max() <= 1 and (d > 0).mean() < 1e-3
        d1 = s.unit_digests()
        s.schedule = "three-stage"
        for t_ in (s.mid, s.up_real, s.up_mid):
            t_.zero_()
        s.run(steps=1)
        assert s.unit_digests() == d1  # the one-launch step and the three stages write the same bytes
        # the fused schedule never writes `mid`: its digest folds the two 4K outputs only -- whatever `mid` holds (it was
        # digesting uninitialised HBM), and equal to the same fold of the unit schedule's outputs
        s.schedule = "fused"
        for t_ in (s.up_real, s.up_mid):
            t_.zero_()
        s.mid.fill_(0x5A)
        s.run(steps=1)
        df = s.unit_digests()
        s.mid.fill_(0xC3)
        assert s.unit_digests() == df and df != d1
        s.schedule = "unit"
        s.run(steps=1)
        assert s.unit_digests() == d1 and s.unit_digests((None, s.up_real, s.up_mid), s.count) == df
    finally:
        s.close()


@pytest.mark.gpu
def test_cli_stream_digests_do_not_depend_on_the_sharding(nsc):
    """`python -m nu_scaler_amd.cli stream` at 1080p on the GPU box: 12 units by one rank, by two and by three ranks (gloo
    rehearsal, all on device 0: the box has one GPU) -- the same 12 digests, in stream order, whoever computed them."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)

    def run(gpus, units, pattern):
        cmd = [sys.executable, "-m", "nu_scaler_amd.cli", "stream", "--gpus", str(gpus), "--units", str(units), "--steps", "2",
               "--pattern", pattern, "--digest", "--backend", "gloo", "--force-device", "0"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return lines[0]

    for pattern in ("gradient", "noise"):
        one = run(1, 12, pattern)
        two = run(2, 6, pattern)
        three = run(3, 4, pattern)
        assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and three["n_gpus"] == 3
        assert two["units_by_rank"] == [6, 6] and three["first_unit_by_rank"] == [0, 4, 8]
        assert len(one["unit_digests"]) == 12 and one["unit_digests"] == two["unit_digests"] == three["unit_digests"]
        # and in windows (one pass, the next window fetched on a side stream beside the current one's kernels): the same stream
        cmd = [sys.executable, "-m", "nu_scaler_amd.cli", "stream", "--gpus", "2", "--units", "6", "--pattern", pattern, "--digest",
               "--backend", "gloo", "--force-device", "0", "--window", "4"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-3000:]
        win = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")][0]
        assert win["window"] == 4 and win["steps"] == 1 and win["unit_digests"] == one["unit_digests"]
        assert two["lut_bytes"] > 0 and two["backend"] == "gloo" and one["mpix_per_s"] > 0
