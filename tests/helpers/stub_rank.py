"""Stand-in for bench.py's worker under the launcher test: one rank of a gloo job on the CPU.
Rank 0 prints ONE JSON line with what the launcher must relay."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    import nu_scaler_amd as nsc

    # as the product creates its group (ShardedStream): whatever the backend prints while it connects -- RCCL's version banner,
    # gloo's "[Gloo] Rank r is connected to n peer ranks" -- goes to stderr; the job's stdout is for rank 0's ONE JSON line
    with nsc.stream._StdoutToStderr():
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    try:
        blob = nsc.build_tables_blob(64, 36, 128, 72) if rank == 0 else b""
        got = nsc.broadcast_blob(blob, src=0)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        start, count = nsc.shard_frames(10 * world, world, rank)
        # what bench.py's worker gathers about every rank (one all_gather of a float64 vector; None travels as NaN)
        import bench
        from nu_scaler_amd import placement

        place = placement.bind_rank(0, world, apply=False, slot=rank)
        rows = bench.gather_rows({"elapsed_s": 1.0 + rank, "check_ok": 1.0, "sclk_MHz": None if rank else 2100.0,
                                  "cpus_per_rank": place["cpus_per_rank"], "first_frame": start}, world, dist, torch,
                                 torch.device("cpu"))
        if "--fail" in sys.argv and rank == world - 1:
            raise SystemExit(3)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "max": float(t.item()), "lut": len(got), "argv": sys.argv[1:],
                              "shard": [start, count], "rows": rows, "elapsed": bench.spread(rows, "elapsed_s"),
                              "sclk": bench.spread(rows, "sclk_MHz", 0)}), flush=True)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
