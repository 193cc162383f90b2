"""ad-hoc: GPU wall time of the bottom-up solve of test/test6.input (config 1; input text from the golden fixture)"""
import sys
import time

import numpy as np

from multigrid_amd import cycle, lib, mgio

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dt = np.float32 if prec == "f32" else np.float64
g = np.load(f"tests/golden/test6_{prec}.npz")
for kernel in (0, 4):
    for rep in range(3):
        IO = mgio.MgIO(text=str(g["input"]))
        top = mgio.iomg_read(IO, dtype=dt)
        s = cycle.Solver(top)
        s.ctx.set_kernel(kernel)
        s.ctx.sync()
        t0 = time.perf_counter()
        s.mg_gs_cds(mgio.read_method(IO))
        s.ctx.sync()
        t1 = time.perf_counter()
        its = [l.itt for l in top.levels()]
        s.close()
    print(f"test6 {prec} kernel {kernel}: {1e3 * (t1 - t0):.2f} ms, sweeps per level {its}", flush=True)
