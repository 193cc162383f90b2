"""ad-hoc: host time of one SlabLevel.sweep (two launches + building the four plane transfers) with the transfers
themselves stubbed out, on a level small enough that the GPU never is the bottleneck"""
import os
import time

import torch
import torch.distributed as dist

import bench
from multigrid_amd import lib, mgio, slab

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29561")
dist.init_process_group("gloo", rank=0, world_size=1)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = lib.Context(0, st.cuda_stream)
top = mgio.iomg_read(mgio.MgIO(text=bench.config3_input((64, 64, 192))))
sl = slab.SlabLevel(ctx, top, 1, 3)          # the middle rank of three: four transfers per sweep
sl._ordered = True                           # as with RCCL: no host wait before a transfer is posted
built = []
dist.batch_isend_irecv = lambda ops: (built.append(len(ops)), [])[1]
sl.lv.setup(True)
for reps in (20, 2000):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        sl.sweep()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
print(f"host {1e6 * (t1 - t0) / reps:.1f} us per sweep (queue drained {1e3 * (t2 - t1):.2f} ms later), {built[-1]} transfers built per sweep")
