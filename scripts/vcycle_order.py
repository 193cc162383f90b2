"""ad-hoc: the v-cycle solve of the bench input (config 3) at a given size, sor and sweep order, with a heartbeat on
stderr: vcycle_order.py N SOR KERNEL [TOL]   (KERNEL 0 = red-black, 4 = the reference's order)"""
import sys
import threading
import time

import numpy as np

import bench
from multigrid_amd import cycle, mgio

n, sor, kernel = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
tol = sys.argv[4] if len(sys.argv) > 4 else "1.e-8"
top = mgio.iomg_read(mgio.MgIO(text=bench.config3_input((n, n, n), tol=tol, sor=sor)))
s = cycle.Solver(top, print_sums=False)
s.ctx.set_kernel(kernel)
s.max_sweeps = 400000
t0 = time.perf_counter()
stop = threading.Event()


def beat():
    while not stop.wait(30.0):
        print(f"  ... {time.perf_counter() - t0:6.0f} s, sweeps per level {[int(g.itt) for g in top.levels()]}", file=sys.stderr, flush=True)


threading.Thread(target=beat, daemon=True).start()
status = "converged"
try:
    s.mg_gs_cds(cycle.CDS_W_CYCLE)
except Exception as e:
    status = f"stopped: {e}"
s.ctx.sync()
stop.set()
print(f"n={n} sor={sor} kernel={kernel} tol={tol}: {status}, {time.perf_counter() - t0:.2f} s, sweeps per level "
      f"{[int(g.itt) for g in top.levels()]}", flush=True)
s.close()
