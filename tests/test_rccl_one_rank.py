"""RCCL on the GPU box: the path's only collective -- the broadcast of the
filter tables -- executed once through the nccl (= RCCL) backend before the driver's 8-GPU run.

A 1-GPU box cannot hold two RCCL ranks, so ONE rank is started as a fresh child process tree
(`python -m torch.distributed.run --nproc-per-node 1`, rendezvous on 127.0.0.1) and the world-of-one
early return of `broadcast_blob` / `broadcast_tables` is bypassed with `force=True`: communicator
creation and both broadcasts run in librccl on the GPU, the tables that came back feed an upscale
that is compared with the oracle."""
import json
import os
import sys

import pytest

from conftest import ROOT

RANK = os.path.join(ROOT, "tests", "helpers", "rccl_rank.py")


@pytest.mark.gpu
def test_rccl_broadcast_of_tables_one_rank(nsc):
    sys.path.insert(0, ROOT)
    import bench

    rc, lines = bench.launch_ranks(1, RANK, [], timeout=600)
    out = [json.loads(ln) for ln in lines if ln.startswith("{")]
    assert rc == 0 and len(out) == 1, lines
    r = out[0]
    assert r["world"] == 1 and r["backend"] == "nccl"
    assert r["blob_bytes"] == len(nsc.build_tables_blob(1920, 1080, 3840, 2160)) == r["lut_bytes"]
    assert r["blob_identical"], "the blob that came back through RCCL differs from the one sent"
    assert r["rccl_libraries"], "librccl is not mapped in the rank: the collective did not go through RCCL"
    # Lanczos-3, FMA mode: every sample within 1 LSB, fewer than 0.1 % of the samples different
    assert r["max_abs_diff"] <= 1 and r["frac_differing"] < 1e-3, r
    assert r["kernel_variant"].startswith("lanczos3_x2")


@pytest.mark.gpu
def test_sharded_stream_over_rccl_one_rank(nsc):
    """The product's rank loop (nu_scaler_amd.run_sharded) with backend "nccl" on the GPU box: one rank, but every collective of the
    8-GPU job issued on its RCCL communicator (force_collectives) -- communicator creation with device_id, the LUT broadcast, the
    barriers with device_ids around the timed steps, the all_gathers of the per-rank rows on the GPU -- and the shard's outputs
    against the oracle."""
    sys.path.insert(0, ROOT)
    from nu_scaler_amd import launch

    rc, lines = launch.launch_ranks(1, os.path.join(ROOT, "tests", "helpers", "rccl_stream_rank.py"), [], timeout=600)
    out = [json.loads(ln) for ln in lines if ln.startswith("{")]
    assert rc == 0 and len(out) == 1, lines
    r = out[0]
    assert r["n_gpus"] == 1 and r["backend"] is None and r["units_by_rank"] == [6] and r["steps"] == 3
    assert r["lut_bytes"] == len(nsc.build_tables_blob(1920, 1080, 3840, 2160)), "the tables did not go through the broadcast"
    assert r["bad"] == [0] and r["rccl_mapped"] == [True]
    assert r["mpix_per_s"] > 0 and r["placement"]["gpu_bdf_verified"] in (True, False, None)


def test_force_flag_is_a_no_op_without_a_process_group(nsc):
    """CPU: outside torch.distributed both helpers stay local whatever `force` says."""
    blob = nsc.build_tables_blob(64, 36, 128, 72)
    assert nsc.broadcast_blob(blob, 0, None, force=True) == blob
