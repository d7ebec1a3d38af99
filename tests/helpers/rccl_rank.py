"""One rank of an RCCL (torch.distributed backend "nccl") job on a real GPU, started by
tests/test_rccl_one_rank.py through `python -m torch.distributed.run`.

What SURVEY.md section 8(e) has as its only collective -- the broadcast of the filter-table blob --
goes through RCCL here even in a world of one (`force=True`): communicator creation, the size
broadcast and the payload broadcast all execute on the GPU.  The tables that came back are
imported into a second upscaler, which then upscales device-resident frames; rank 0 checks the
result against the CPU oracle and prints ONE JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import numpy as np

        import nu_scaler_amd as nsc

        fetch = nsc.transfer.to_numpy  # device -> host through nus_download, never Tensor.cpu() (docs/d2h_fault_analysis.md)
        import oracle

        oracle.build()
        w, h, n = 1920, 1080, 2  # the BASELINE size: 888 040 bytes of tables
        src = nsc.PyWgpuUpscaler("quality", "lanczos3", device=local)
        src.initialize(w, h, 2 * w, 2 * h)
        sent = src.export_tables() if rank == 0 else b""
        got = nsc.broadcast_blob(sent, 0, dev, force=True)  # RCCL: int64 size, then the payload
        nsc.validate_tables_blob(got, w, h, 2 * w, 2 * h)
        dst = nsc.PyWgpuUpscaler("quality", "lanczos3", device=local)
        dst.initialize(w, h, 2 * w, 2 * h)
        dst.import_tables(got)
        lut = nsc.broadcast_tables(dst, 0, dev, force=True)  # the bench's own call, world-1 return bypassed
        frames = nsc.transfer.to_device(np.stack([oracle.gen_noise(w, h, 11 + rank * n + k) for k in range(n)]), dev)
        out = torch.empty((n, 2 * h, 2 * w, 4), dtype=torch.uint8, device=dev)
        dst.upscale_device(frames.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        worst, differing = 0, 0.0
        for k in range(n):
            want = oracle.lanczos3(fetch(frames[k]), 2 * w, 2 * h, threads=0).astype(np.int16)
            d = np.abs(fetch(out[k]).astype(np.int16) - want)
            worst, differing = max(worst, int(d.max())), max(differing, float((d != 0).mean()))
        t = torch.tensor([float(worst)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # one more collective kind through RCCL
        with open("/proc/self/maps") as f:
            rccl = sorted({ln.split()[-1] for ln in f if "librccl" in ln})
        if rank == 0:
            print(json.dumps({"world": world, "backend": dist.get_backend(), "blob_bytes": len(got),
                              "blob_identical": got == sent, "lut_bytes": lut, "max_abs_diff": int(t.item()),
                              "frac_differing": differing, "rccl_libraries": rccl,
                              "kernel_variant": dst.kernel_variant}), flush=True)
        dist.barrier(device_ids=[local])
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
