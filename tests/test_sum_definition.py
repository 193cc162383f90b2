"""What the order-independent sum changes (CPU only, oracle in red-black order).

The reference's grid_sum (t_mg.f90:794-814) accumulates |V| sequentially in real(grid_p); the GPU path and the
oracle's SUM_EXACT define the same quantity as an exact integer sum rounded once.  The convergence tests
(m_gs_CDS.F90:150, 207) compare differences of such sums with a tolerance, so the definition can only matter
through rounding.  In fp64 it does not: on the reference's own solve inputs both sums give identical sweep counts
per level, identical fields, and histories within 1e-12.  In fp32 (the precision the reference ships) the
sequential sum carries a rounding error of ~1e-4 relative at these sizes, so the two definitions differ at that
level and a convergence test may trip one sweep earlier or later; that bound is what the fp32 cases assert.
"""
import numpy as np
import pytest

from multigrid_amd import mgio
from oracle import oracle as orc
from tests.helpers import DT, load, oracle_levels

# solve_80_v is run at sor 1.6 (red-black order does not converge on it at the reference's 1.8)
CASES = [("solve_64_bu", None), ("solve_80_bu", None), ("solve_64x48x80_v_half", None), ("test6", None),
         ("solve_80_v", ("sor 1.8", "sor 1.6"))]


def both_sums(name, swap, prec):
    text = str(load(f"{name}_{prec}")["input"])
    if swap:
        assert swap[0] in text
        text = text.replace(*swap)
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=DT[prec])
    method = mgio.read_method(IO)
    out = []
    for mode in (orc.SUM_SEQ, orc.SUM_EXACT):
        out.append(orc.solve(oracle_levels(top), method, DT[prec], order=orc.ORDER_RB, sum_mode=mode, max_sweeps=20000))
    return out


@pytest.mark.parametrize("name,swap", CASES)
def test_fp64_sequential_and_exact_sum_give_the_same_solve(name, swap):
    (V0, h0, p0), (V1, h1, p1) = both_sums(name, swap, "f64")
    assert p0 == p1                                   # identical sweep counts on every level
    assert len(h0) == len(h1)
    for f in ("layer", "kind", "itt"):
        assert np.array_equal(h0[f], h1[f])
    assert np.array_equal(V0, V1)                     # same decisions, same sweeps: the same field bit for bit
    assert np.array_equal(h0["err"], h1["err"])
    rel = np.abs(h0["sum"] - h1["sum"]) / np.abs(h1["sum"])
    assert rel.max() <= 1e-12, rel.max()
    # tol = |old_sum - new_sum| * nr: a difference of sums, compared absolutely (sums are O(1) per cell here)
    assert np.abs(h0["tol"] - h1["tol"]).max() <= 1e-12


@pytest.mark.parametrize("name,swap", CASES[:4])
def test_fp32_sums_differ_by_rounding_of_the_sequential_sum(name, swap):
    (V0, h0, p0), (V1, h1, p1) = both_sums(name, swap, "f32")
    # the sequential fp32 sum of up to 8.6e6 terms is off by up to ~1e-4 relative; a test that compares
    # differences of sums with tol * nr can therefore trip a few sweeps apart
    assert len(p0) == len(p1)
    for a, b in zip(p0, p1):
        assert abs(a - b) <= max(4, 0.02 * max(a, b)), (p0, p1)
    n = min(len(h0), len(h1))
    same = (h0["layer"][:n] == h1["layer"][:n]) & (h0["itt"][:n] == h1["itt"][:n])
    k = int(np.argmin(same)) if not same.all() else n   # compare the common prefix of the two histories
    if k:
        rel = np.abs(h0["sum"][:k] - h1["sum"][:k]) / np.abs(h1["sum"][:k])
        assert rel.max() <= 1e-3, rel.max()
