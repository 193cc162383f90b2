"""ad-hoc: the undivided 256 x 256 x 512 level of the 2-rank bench rehearsal: 12 sweeps, print the reductions"""
import numpy as np
import bench
from multigrid_amd import cycle, lib, mgio
top = mgio.iomg_read(mgio.MgIO(text=bench.config3_input((256, 256, 512), sor="1.7")))
s = cycle.Solver(top)
s.bring_back(top); s.setup(top, True)
for _ in range(12):
    top.dev.sweep(1, 3)
print(top.dev.result())
