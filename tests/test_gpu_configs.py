"""GPU: the benchmark configurations that are parity cases rather than bench lines, at sizes the oracle solves in
seconds.  Config 1 is the reference's own test/test6.input (240 x 150 x 240, odd coarse sizes 75 -> 37, a per-layer
tolerance; fp32 as shipped and fp64); config 4 is the elongated multi-GPU case scaled to 40 x 40 x 640 (x and y stop
halving, four levels).  Config 2 (all periodic, v-cycle) is solve_80_v of tests/test_gpu_solve.py; config 3 at full
size is tests/test_gpu_full_size.py.  Bar: residual history, sweeps per level and the final potential bit for bit
against the oracle's red-black solve."""
import numpy as np
import pytest

from multigrid_amd import cycle, mgio
from oracle import oracle as orc
from tests.helpers import DT, bits_equal, config4_input, load, oracle_levels

pytestmark = pytest.mark.gpu


def _solve_and_compare(text, dtype):
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=dtype)
    method = mgio.read_method(IO)
    V, hist, per = orc.solve(oracle_levels(top), method, dtype, order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
    s = cycle.Solver(top)
    try:
        s.mg_gs_cds(method)
        h = s.history_array()
        assert [l.itt for l in top.levels()] == per
        assert len(h) == len(hist)
        for f in ("layer", "kind", "itt", "tol", "sum", "err"):
            assert np.array_equal(h[f], hist[f]), f
        assert bits_equal(top.dev.download(), V)
    finally:
        s.close()
    return top, per


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_config1_test6_input(prec):
    top, per = _solve_and_compare(str(load(f"test6_{prec}")["input"]), DT[prec])
    assert [tuple(int(v) for v in g.n) for g in top.levels()] == [(240, 150, 240), (120, 75, 120), (60, 37, 60)]
    assert all(k > 0 for k in per)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_config4_elongated(prec):
    top, per = _solve_and_compare(config4_input(), DT[prec])
    assert [tuple(int(v) for v in g.n) for g in top.levels()] == [(40, 40, 640), (20, 20, 320), (20, 20, 160), (20, 20, 80)]
    assert [g.tol for g in top.levels()] == [1e-3, 1e-3, 1e-3, 1e-4]


def test_mg_program_runs_test6_input(tmp_path, monkeypatch):
    """`python -m multigrid_amd <input>`: the reference's mg program on its own test input, fp32 as shipped; the saved
    .VMG holds the oracle's potential (planes 1..n3 with their x/y ghosts)"""
    from multigrid_amd import __main__ as mg
    text = str(load("test6_f32")["input"])
    (tmp_path / "t6.input").write_text(text)
    monkeypatch.chdir(tmp_path)
    assert mg.main([str(tmp_path / "t6.input")]) == 0
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=np.float32)
    V, _, _ = orc.solve(oracle_levels(top), mgio.read_method(IO), np.float32, order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
    n, offset, cell, planes = mgio.read_vmg(str(tmp_path / "out.VMG"), np.float32)
    assert [int(v) for v in n] == [240, 150, 240]
    assert bits_equal(planes, V[1:-1])
