"""shared helpers for the tests: fixture loading and oracle Level construction"""
import os

import numpy as np

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f32": np.float32, "f64": np.float64}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def boxes_of(g):
    return [orc.Box(tuple(int(v) for v in p), float(v), float(r), bool(c))
            for p, v, r, c in zip(g["box_place"], g["box_val"], g["box_rho"], g["box_const"])]


def level_of(g, n_key="n", boxes=None, **kw):
    a = tuple(float(x) for x in g["a"]) if "a" in g.files else (-1, -1, -1)
    if a[0] < 0:
        a = (1.0 / 3.0,) * 3          # init_grid default, t_mg.f90:143
    return orc.Level(n=tuple(int(v) for v in g[n_key]), bc=tuple(int(v) for v in g["bc"]),
                     sor=float(g["sor"]) if "sor" in g.files else 1.8, a=a,
                     boxes=boxes_of(g) if boxes is None else boxes, **kw)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    iv = np.uint32 if a.dtype == np.float32 else np.uint64
    return np.array_equal(a.view(iv), b.view(iv))


def oracle_levels(top):
    """oracle Level list from a multigrid_amd.grid.MgGrid hierarchy"""
    out = []
    for g in top.levels():
        out.append(orc.Level(
            n=tuple(int(v) for v in g.n), bc=tuple(g.bc6()), sor=float(g.sor), a=tuple(float(v) for v in g.a),
            tol=float(g.tol), steps=int(g.steps), res_method=int(g.RES_method), pro_method=int(g.PRO_method),
            boxes=[orc.Box(tuple(int(max(-2**31, min(2**31 - 1, v))) for v in b.place), float(b.val), float(b.rho),
                           bool(b.constant)) for b in g.box]))
    return out


def config4_input(nx=40, nz=640, tol="1.e-3", tol_last="1.e-4", method="v-cycle"):
    """the elongated multi-GPU configuration (256 x 256 x 4096 in full) scaled down: electrodes at the two z ends,
    x and y stop halving while z goes on (t_mg.f90:191-194), `max-layers 4`, a per-layer block for the coarsest level"""
    Lx, Lz = nx * 0.1, nz * 0.1
    return f"""
cell {Lx:.10g} {Lx:.10g} {Lz:.10g}
max-layers 4
boxes 4
method {method}
begin layers
  sor 1.7
  tol {tol}
  v-steps 2
  mesh-size {nx} {nx} {nz}
  bc all periodic
  bc c- neumann
  bc c+ dirichlet
  interp full
end layers
begin layer -1
  tol {tol_last}
end layer
begin box
  llc {0.25 * Lx:.10g} {0.25 * Lx:.10g} 0.
  cell {0.5 * Lx:.10g} {0.5 * Lx:.10g} {0.05 * Lz:.10g}
  value 0.5
end box
begin box
  llc {0.25 * Lx:.10g} {0.25 * Lx:.10g} {0.95 * Lz:.10g}
  cell {0.5 * Lx:.10g} {0.5 * Lx:.10g} {0.05 * Lz:.10g}
  value -0.5
end box
"""
