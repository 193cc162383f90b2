"""On the GPU box: wall time of whole solves of small inputs (default: the reference's test/test6.input as kept in
tests/golden), second run of two, with the sweeps per level; MGX_SOLVE_LOOP=0 selects a launch per sweep in
mgx_level_solve.   python scripts/time_small_solve.py [name] [f32|f64]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigrid_amd import cycle, mgio          # noqa: E402
from tests.helpers import DT, load              # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "test6"
prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
text = str(load(f"{name}_{prec}")["input"])
best = None
for rep in range(3):
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=DT[prec])
    method = mgio.read_method(IO)
    s = cycle.Solver(top, print_sums=False)
    s.ctx.sync()
    t0 = time.perf_counter()
    s.mg_gs_cds(method)
    s.ctx.sync()
    dt = time.perf_counter() - t0
    per = [g.itt for g in top.levels()]
    chk = float(np.abs(top.dev.download()).sum())
    s.close()
    best = dt if best is None else min(best, dt)
print(f"{name} {prec} MGX_SOLVE_LOOP={os.environ.get('MGX_SOLVE_LOOP', '(default)')}: best of 3 {best * 1e3:.2f} ms, sweeps {per}, sum|V| {chk!r}")
