import sys
import numpy as np
from multigrid_amd import cycle, mgio, lib
from oracle import oracle as orc
from tests.helpers import DT, load, oracle_levels

name, prec = sys.argv[1], sys.argv[2]
g = load(f"{name}_{prec}")
IO = mgio.MgIO(text=str(g["input"]))
top = mgio.iomg_read(IO, dtype=DT[prec])
method = mgio.read_method(IO)
V, hist, per = orc.solve(oracle_levels(top), method, DT[prec], order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
print("oracle per", per, "nhist", len(hist))
print(hist[:6])
s = cycle.Solver(top, verbose=False)
s.ctx.set_kernel(1)
try:
    s.mg_gs_cds(method)
except Exception as e:
    print("EXC", e)
h = s.history_array()
print("gpu nhist", len(h))
print(h[:6])
k = min(len(h), len(hist))
bad = [i for i in range(k) if h[i]["sum"] != hist[i]["sum"] or h[i]["err"] != hist[i]["err"]]
print("first mismatches", bad[:5])
for i in bad[:3]:
    print(i, h[i], hist[i])
for l in top.levels():
    try:
        W = l.dev.download()
        print("layer", l.layer, "finite", np.isfinite(W).all(), "absmax", np.abs(W).max(), "itt", l.itt)
    except Exception as e:
        print("layer", l.layer, "download failed:", e)
