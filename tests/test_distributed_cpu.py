"""CPU, world_size 2, gloo: the multi-GPU path's host logic -- contiguous frame shards
with an overlap frame, and the one-off table broadcast (RCCL on the GPU box)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nu_scaler_amd as nsc
        from nu_scaler_amd import synthetic as syn

        # 1. table broadcast: rank 0 builds, everyone receives the identical, valid blob
        mine = nsc.build_tables_blob(64, 36, 128, 72)
        got = nsc.broadcast_blob(mine if rank == 0 else b"", src=0)
        nsc.validate_tables_blob(got, 64, 36, 128, 72)
        same = got == mine
        # 2. frame sharding: each rank materialises its chunk (+ overlap frame) of the stream
        n_units = 11
        start, count = nsc.shard_frames(n_units, world, rank)
        frames = syn.gradient_stream_torch(count + 1, 32, 8, "cpu", first=start)
        pairs = [(start + i, start + i + 1) for i in range(count)]
        digest = torch.tensor([int(frames[i].sum()) for i in range(count)] + [0] * (n_units - count), dtype=torch.int64)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([count], dtype=torch.int64))
        dist.barrier()
        q.put((rank, same, start, count, pairs, [int(c) for c in counts], digest[:count].tolist()))
    finally:
        dist.destroy_process_group()


def test_world2_shards_and_table_broadcast(nsc):
    from nu_scaler_amd import synthetic as syn

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "broadcast blob differs from the locally built one"
    all_pairs = [pr for r in res for pr in r[4]]
    assert all_pairs == [(k, k + 1) for k in range(11)], "every (k, k+1) pair exactly once, in stream order"
    assert res[0][5] == [6, 5] and res[1][5] == [6, 5]
    want = [int(syn.gradient_frame(32, 8, k).astype(np.int64).sum()) for k in range(11)]
    assert res[0][6] + res[1][6] == want
