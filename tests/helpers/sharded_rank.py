"""One rank of tests/test_sharded_stream.py's CPU jobs: nu_scaler_amd.stream.run_sharded on HOST tensors over gloo, with an
oracle-backed stand-in for the pipeline (test infrastructure: the product has no CPU path) whose `upscaler` stand-in carries the
real table blob through the real broadcast.  Every rank checks every unit of its shard against the oracle; rank 0 prints ONE
JSON line."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

W, H = 48, 20


class _Tables:
    """What broadcast_tables needs of an upscaler: rank 0 exports the blob the real upscaler of these dimensions would export
    (built host-only through the C ABI), the others validate and keep what arrives."""

    def __init__(self, nsc):
        self.nsc, self.imported = nsc, 0

    def export_tables(self):
        return self.nsc.build_tables_blob(W, H, 2 * W, 2 * H)

    def import_tables(self, blob):
        self.nsc.validate_tables_blob(blob, W, H, 2 * W, 2 * H)
        assert blob == self.export_tables()
        self.imported += 1


class HostPipeline:
    def __init__(self, w, h, device_index, stream, nsc, oracle):
        self.w, self.h, self.oracle = w, h, oracle
        self.upscaler = _Tables(nsc)
        self.unit_pixels = 3 * w * h + 2 * (w * h + 4 * w * h)
        self.unit_bytes = self.unit_pixels * 4

    def alloc(self, n, device):
        return (torch.zeros((n, self.h, self.w, 4), dtype=torch.uint8), torch.zeros((n, 2 * self.h, 2 * self.w, 4), dtype=torch.uint8),
                torch.zeros((n, 2 * self.h, 2 * self.w, 4), dtype=torch.uint8))

    def step_unit(self, frames, mid, up_real, up_mid, stream=0):
        o = self.oracle
        for k in range(mid.shape[0]):
            a, b = frames[k].numpy(), frames[k + 1].numpy()
            m = o.warp_blend(a, b, None, 0.5)
            mid[k] = torch.from_numpy(m)
            up_real[k] = torch.from_numpy(o.lanczos3(a, 2 * self.w, 2 * self.h))
            up_mid[k] = torch.from_numpy(o.lanczos3(m, 2 * self.w, 2 * self.h))

    step = step_unit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units-total", type=int, default=7)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--sink-fails-on", type=int, default=-1, help="the sink raises on this rank (the job must end, not hang)")
    ap.add_argument("--window", type=int, default=0, help="> 0: the shard in windows of this many units (ShardedStream.run_windows)")
    args = ap.parse_args()
    import nu_scaler_amd as nsc
    import oracle

    oracle.build()
    pipes = []

    def factory(w, h, device_index, stream):
        pipes.append(HostPipeline(w, h, device_index, stream, nsc, oracle))
        return pipes[-1]

    def sink(s):
        if s.rank == args.sink_fails_on:
            raise ValueError("sink failed on purpose")
        # every unit of THIS rank's shard, regenerated from its position in the global stream
        bad = 0
        for k in range(s.count):
            a, b = oracle.gen_gradient(W, H, s.start + k), oracle.gen_gradient(W, H, s.start + k + 1)
            m = oracle.warp_blend(a, b, None, 0.5)
            ok = (np.array_equal(s.frames[k].numpy(), a) and np.array_equal(s.frames[k + 1].numpy(), b)
                  and np.array_equal(s.mid[k].numpy(), m)
                  and np.array_equal(s.up_real[k].numpy(), oracle.lanczos3(a, 2 * W, 2 * H))
                  and np.array_equal(s.up_mid[k].numpy(), oracle.lanczos3(m, 2 * W, 2 * H)))
            bad += 0 if ok else 1
        per_rank = -(-s.total_units // s.world)
        d = s.unit_digests()
        row = {"checked": float(s.count), "bad": float(bad), "imported": float(pipes[0].upscaler.imported)}
        row.update({f"digest_{k:05d}": (float(d[k]) if k < len(d) else None) for k in range(per_rank)})
        return row

    if args.window > 0:
        return windows_main(args, nsc, oracle, factory)
    # the process group is the caller's here (ShardedStream creates one only if none exists, and then tears it down itself;
    # tests/test_sharded_stream.py's GPU tests and the CLI go that way): it outlives run_sharded for the last assertion
    import torch.distributed as dist

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env > 1:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world_env)
    out = nsc.run_sharded(args.units_total, W, H, steps=args.steps, warmup=args.warmup, sink=sink, backend="gloo",
                          device_kind="cpu", pipeline_factory=factory)
    world, rank = out["n_gpus"], out["rank"]
    rows = out.pop("rows")
    # the summary must be the same on every rank
    key = json.dumps({k: out[k] for k in ("units_per_s", "mpix_per_s", "elapsed_s", "units_by_rank", "first_unit_by_rank")}, sort_keys=True)
    same = True
    if world > 1:
        assert dist.is_initialized(), "run_sharded tore down a process group it did not create"
        got = [None] * world
        dist.all_gather_object(got, key)
        same = all(g == got[0] for g in got)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        digests = []
        for r in rows:
            digests += [int(r[k]) for k in sorted(r) if k.startswith("sink_digest_") and r[k] is not None]
        out.update(checked_units_by_rank=[int(r["sink_checked"]) for r in rows], mismatches_by_rank=[int(r["sink_bad"]) for r in rows],
                   tables_imported_by_rank=[int(r["sink_imported"]) for r in rows], unit_digests=digests,
                   summary_equal_on_all_ranks=same, unit_pixels=pipes[0].unit_pixels)
        out.pop("placement", None)
        print(json.dumps(out), flush=True)


def windows_main(args, nsc, oracle, factory):
    """The same stream in windows: every unit of every window against the oracle, digests gathered as the CLI gathers them."""
    s = nsc.ShardedStream(args.units_total, W, H, backend="gloo", device_kind="cpu", pipeline_factory=factory, resident=False)
    try:
        digests, bad, seen = [], [0], []

        def consume(st, first, n, mid, up_real, up_mid):
            seen.append((first, n))
            for k in range(n):
                a, b = oracle.gen_gradient(W, H, first + k), oracle.gen_gradient(W, H, first + k + 1)
                m = oracle.warp_blend(a, b, None, 0.5)
                ok = (np.array_equal(mid[k].numpy(), m) and np.array_equal(up_real[k].numpy(), oracle.lanczos3(a, 2 * W, 2 * H))
                      and np.array_equal(up_mid[k].numpy(), oracle.lanczos3(m, 2 * W, 2 * H)))
                bad[0] += 0 if ok else 1
            digests.extend(st.unit_digests((mid, up_real, up_mid), n))

        s.run_windows(args.window, consume)
        per_rank = -(-s.total_units // s.world)
        row = {"elapsed_s": s.elapsed_local, "first_unit": float(s.start), "units": float(s.count), "bad": float(bad[0]),
               "windows": float(len(seen))}
        row.update({f"digest_{k:05d}": (float(digests[k]) if k < len(digests) else None) for k in range(per_rank)})
        rows = s.gather(row)
        if s.rank == 0:
            out = s.summarize(rows)
            allv = []
            for r in rows:
                allv += [int(r[k]) for k in sorted(r) if k.startswith("digest_") and r[k] is not None]
            out.update(unit_digests=allv, mismatches_by_rank=[int(r["bad"]) for r in rows], windows_by_rank=[int(r["windows"]) for r in rows])
            print(json.dumps(out), flush=True)
    finally:
        s.close()


if __name__ == "__main__":
    main()
