"""One rank of the product's sharded stream over RCCL on a real GPU (tests/test_rccl_one_rank.py, started through
`python -m torch.distributed.run --nproc-per-node 1`): nu_scaler_amd.run_sharded with backend "nccl" and force_collectives=True --
a world of one still creates its RCCL communicator (device_id = its GPU) and issues on it EVERY collective the 8-GPU job will:
the LUT broadcast (size + payload), the barriers around the timed steps (device_ids=[...]), the all_gathers of the per-rank rows
(float64 on the GPU).  The sink checks units of the shard against the oracle; rank 0 prints ONE JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import nu_scaler_amd as nsc

    fetch = nsc.transfer.to_numpy  # device -> host through nus_download, never Tensor.cpu() (docs/d2h_fault_analysis.md)
    import oracle

    oracle.build()
    w, h, n = 1920, 1080, 6

    def sink(s):
        import torch.distributed as dist

        assert dist.is_initialized() and dist.get_backend() == "nccl"
        bad = 0
        for k in (0, s.count - 1):
            a, b = oracle.gen_gradient(w, h, s.start + k), oracle.gen_gradient(w, h, s.start + k + 1)
            m = oracle.warp_blend(a, b, None, 0.5, threads=0)
            ok = np.array_equal(fetch(s.mid[k]), m)
            for got, src in ((s.up_real, a), (s.up_mid, m)):
                d = np.abs(fetch(got[k]).astype(np.int16) - oracle.lanczos3(src, 2 * w, 2 * h, threads=0).astype(np.int16))
                ok = ok and d.max() <= 1 and (d > 0).mean() < 1e-3
            bad += 0 if ok else 1
        with open("/proc/self/maps") as f:
            rccl = any("librccl" in ln for ln in f)
        return {"bad": float(bad), "rccl_mapped": 1.0 if rccl else 0.0}

    out = nsc.run_sharded(n, w, h, steps=3, warmup=1, sink=sink, backend="nccl", force_collectives=True)
    if out["rank"] == 0:
        rows = out.pop("rows")
        out.update(bad=[int(r["sink_bad"]) for r in rows], rccl_mapped=[bool(r["sink_rccl_mapped"]) for r in rows])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
