"""GPU parity of whole solves: the host cycle control over the HIP library vs the oracle's
red-black / exact-sum solve of the same input.  Bar: per-level potential bit-exact, residual
history bit-exact, identical sweep counts (the requirement is <= 1e-12 relative)."""
import numpy as np
import pytest

from multigrid_amd import cycle, mgio
from oracle import oracle as orc
from tests.helpers import DT, bits_equal, load, oracle_levels

pytestmark = pytest.mark.gpu

SOLVES = ["solve_64_bu", "solve_80_bu", "solve_80_v", "solve_64x48x80_v_half"]


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", SOLVES)
def test_solve_matches_oracle(name, prec, kernel):
    g = load(f"{name}_{prec}")
    IO = mgio.MgIO(text=str(g["input"]))
    top = mgio.iomg_read(IO, dtype=DT[prec])
    method = mgio.read_method(IO)
    V, hist, per = orc.solve(oracle_levels(top), method, DT[prec], order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
    s = cycle.Solver(top)
    try:
        s.ctx.set_kernel(kernel)
        s.mg_gs_cds(method)
        W = top.dev.download()
        h = s.history_array()
        assert [l.itt for l in top.levels()] == per
        assert len(h) == len(hist)
        for f in ("layer", "kind", "itt"):
            assert np.array_equal(h[f], hist[f]), f
        for f in ("tol", "sum", "err"):
            assert np.array_equal(h[f], hist[f]), f
        assert bits_equal(W, V)
    finally:
        s.close()
