"""Worker of tests/test_gpu_dist_solve.py (one process per rank, started by torch.distributed.run): solves an input
with every level decomposed into z slabs and writes the history and the gathered top-level field on rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from multigrid_amd import lib, mgio, slab
    case, out, backend = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = max(1, torch.cuda.device_count())
    dev = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    d = np.load(case, allow_pickle=False)
    dtype = np.dtype(str(d["dtype"]))
    IO = mgio.MgIO(text=str(d["input"]))
    top = mgio.iomg_read(IO, dtype=dtype)
    method = mgio.read_method(IO)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = lib.Context(dev, stream.cuda_stream)
    below = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    s = slab.SlabSolver.build(top, ctx, rank, world, replicate_below=below)
    s.mg_gs_cds(method)
    mine = top.dev.download()[1:-1]                      # owned planes, x/y ghosts included
    parts = [None] * world if rank == 0 else None
    dist.gather_object((top.dev.z0, mine), parts, dst=0)
    if rank == 0:
        planes = np.concatenate([p for _, p in sorted(parts, key=lambda t: t[0])], axis=0)
        h = s.history_array()
        np.savez(out, planes=planes, per=np.array([g.itt for g in top.levels()]),
                 kinds=np.array([type(g.dev).__name__ for g in top.levels()]),
                 depths=np.array([getattr(g.dev, "depth", 0) for g in top.levels()]),
                 **{f"h_{k}": h[k] for k in ("layer", "kind", "itt", "tol", "sum", "err")})
    dist.barrier()
    s.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
