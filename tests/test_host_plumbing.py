"""Host plumbing (no GPU): input grammar + level hierarchy + box index ranges against what the
reference's own `mg` program printed for the same input files (fixtures tests/golden/solve_*.npz,
test6_*.npz), and the oracle's cycle control against the reference's final fields."""
import numpy as np
import pytest

from multigrid_amd import grid as G
from multigrid_amd import mgio
from oracle import oracle as orc
from tests.helpers import DT, bits_equal, load, oracle_levels

SOLVES = ["solve_64_bu", "solve_80_bu", "solve_80_v", "solve_64x48x80_v_half", "test6"]


def parse(g, prec):
    IO = mgio.MgIO(text=str(g["input"]))
    top = mgio.iomg_read(IO, dtype=DT[prec])
    return IO, top


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", SOLVES)
def test_parser_and_boxes_match_reference_print(name, prec):
    g = load(f"{name}_{prec}")
    IO, top = parse(g, prec)
    lv = top.levels()
    assert [list(map(int, l.n)) for l in lv] == g["layer_n"].tolist()
    for i, l in enumerate(lv):
        nb = len(l.box)
        assert nb == g["box_place"].shape[1]
        for j, b in enumerate(l.box):
            assert [int(v) for v in b.place] == g["box_place"][i, j].tolist(), (name, i, j)
            assert int(b.constant) == int(g["box_const"][i, j])
            assert float(f"{b.val:.3E}") == g["box_val_print"][i, j]
        assert float(f"{G.grid_tolerance(l):.3E}") == pytest.approx(g["tolerance_print"][i], rel=2e-3)
    want_method = 2 if len(g["w_counts"]) else 1
    assert mgio.read_method(IO) == want_method
    assert mgio.save_names(IO) == ["out.vmg"]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", ["solve_64_bu", "solve_80_bu", "solve_80_v", "solve_64x48x80_v_half"])
def test_oracle_cycle_control_matches_reference(name, prec):
    """lexicographic sweeps + sequential sums = the reference at one thread: final field bit for bit"""
    g = load(f"{name}_{prec}")
    IO, top = parse(g, prec)
    method = mgio.read_method(IO)
    V, hist, per = orc.solve(oracle_levels(top), method, DT[prec], order=orc.ORDER_LEX, sum_mode=orc.SUM_SEQ)
    assert bits_equal(V[1:-1], g["V"])
    if method == 1:
        want = {int(l): int(c) for l, c in g["bu_counts"]}
        assert {i + 1: per[i] for i in range(len(per))} == want
    else:
        # "Completed: p:c in N itt" accumulates per pg level
        tot = {}
        for p, c, nitt in g["w_counts"]:
            tot[int(p)] = tot.get(int(p), 0) + int(nitt)
        for layer, nitt in tot.items():
            assert per[layer - 1] >= nitt
    # the 7-decimal history the reference printed
    h = g["hist"]
    mine = hist[(hist["kind"] == 0) | (hist["kind"] == 2)]
    assert len(mine) == len(h)
    assert np.allclose(mine["tol"], h[:, 1], atol=6e-8)
    assert np.allclose(mine["err"], h[:, 3], atol=6e-8)


def test_test6_counts_fp64_oracle():
    """config 1 plumbing check: sweep counts of test/test6.input (fp64 variant: 126/9/4)"""
    g = load("test6_f64")
    IO, top = parse(g, "f64")
    V, hist, per = orc.solve(oracle_levels(top), 1, np.float64, order=orc.ORDER_LEX, sum_mode=orc.SUM_SEQ)
    assert per == [int(c) for _, c in sorted(g["bu_counts"].tolist())]
    sub = int(g["subsample"])
    assert bits_equal(np.ascontiguousarray(V[1:-1][::sub, ::sub, ::sub]), np.ascontiguousarray(g["V"]))


def test_children_halving_rule():
    top = G.MgGrid()
    cell = np.diag([25.6, 25.6, 409.6])
    G.init_grid(top, (256, 256, 4096), cell, 2, tol=1e-8)
    G.init_grid_children_half(top)
    assert [tuple(int(v) for v in l.n) for l in top.levels()] == [
        (256, 256, 4096), (128, 128, 2048), (64, 64, 1024), (32, 32, 512), (32, 32, 256), (32, 32, 128), (32, 32, 64)]
    top = G.MgGrid()
    G.init_grid(top, (512, 512, 512), np.diag([51.2] * 3), 2, tol=1e-8)
    G.init_grid_children_half(top, max_layer=3)
    assert [int(l.n[0]) for l in top.levels()] == [512, 256, 128]


def test_vmg_roundtrip(tmp_path):
    top = G.MgGrid(np.float32)
    G.init_grid(top, (5, 4, 3), np.diag([1.0, 2.0, 3.0]), 0, tol=1e-3)
    V = np.arange(5 * 6 * 7, dtype=np.float32).reshape(5, 6, 7)
    p = str(tmp_path / mgio.vmg_name("field.vmg"))
    assert p.endswith("field.VMG")
    mgio.write_vmg(p, top, V)
    n, off, cell, W = mgio.read_vmg(p, np.float32)
    assert n.tolist() == [5, 4, 3] and np.array_equal(W, V[1:-1]) and np.array_equal(cell, top.cell)


def test_reference_test7_index_maps():
    """the reference's test/test7.f90 on its own 200^3 hierarchy (max_layer 4): walking a parent direction, g2g reaches
    every child index and none beyond; walking a child direction, it reaches as many distinct parent indices as the
    child has cells, none beyond the parent"""
    from multigrid_amd import grid as G
    top = G.MgGrid(np.dtype(np.float32))
    cell = np.diag([31.75, 20.0, 31.75])
    G.init_grid(top, [200, 200, 200], cell, 4, tol=1e-4)
    G.init_grid_children_half(top, max_layer=4)
    levels = list(top.levels())
    assert [int(g.n[0]) for g in levels] == [200, 100, 50, 25]
    for g, c in zip(levels[:-1], levels[1:]):
        for i in range(3):
            n, nc = int(g.n[i]), int(c.n[i])
            down = [G.g2g(n, p, nc) for p in range(1, n + 1)]
            assert max(down) <= nc and len(set(down)) == nc
            up = [G.g2g(nc, p, n) for p in range(1, nc + 1)]
            assert max(up) <= n and len(set(up)) == nc
