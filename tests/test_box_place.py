"""grid_add_box (t_mg.f90:323-435): the index range of a box on a level.

CPU: the oracle's literal loop and the host's per-axis arithmetic against what the compiled reference printed
(tests/golden/box_place.npz, made by oracle/make_golden.py boxplace from oracle/_ref/ref_boxplace_f64).
GPU: mgx_box_place through the C ABI against the same fixture, against the oracle on larger random cases, and
the host hierarchy (grid.grid_add_box) on a skewed cell, which has no host form of the loop."""
import numpy as np
import pytest

from multigrid_amd import grid as G
from oracle import oracle as orc
from tests.helpers import load

CASES = ("ortho", "skew", "hexa", "thin")


def _case(g, name):
    n = [int(v) for v in g[f"{name}_n"]]
    cell = g[f"{name}_cell"]
    return n, cell, cell / np.array(n, dtype=np.float64)[None, :], g[f"{name}_offset"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_reference(name):
    g = load("box_place")
    n, cell, dL, off = _case(g, name)
    for llc, bc, want in zip(g[f"{name}_llc"], g[f"{name}_box_cell"], g[f"{name}_place"]):
        assert np.array_equal(orc.box_place(n, dL, off, llc, bc), want)


def _host_places(n, cell, off, llcs, bcs):
    top = G.MgGrid(np.float64)
    G.init_grid(top, n, cell, len(llcs), tol=1e-4, offset=off)
    for llc, bc in zip(llcs, bcs):
        G.grid_add_box(top, llc, bc, 1.0, 1.0, True)
    return np.array([b.place for b in top.box], dtype=np.int64)


def test_host_per_axis_arithmetic_equals_reference():
    g = load("box_place")
    n, cell, dL, off = _case(g, "ortho")
    got = _host_places(n, cell, off, g["ortho_llc"], g["ortho_box_cell"])
    assert np.array_equal(got, g["ortho_place"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_equals_reference(name):
    from multigrid_amd import lib
    g = load("box_place")
    n, cell, dL, off = _case(g, name)
    ctx = lib.shared_context()
    for llc, bc, want in zip(g[f"{name}_llc"], g[f"{name}_box_cell"], g[f"{name}_place"]):
        assert np.array_equal(ctx.box_place(n, dL, off, llc, bc), want), (name, llc)


@pytest.mark.gpu
def test_device_equals_oracle_random():
    from multigrid_amd import lib
    ctx = lib.shared_context()
    rng = np.random.default_rng(7)
    for n in ((96, 80, 72), (1, 1, 1), (130, 3, 65), (64, 64, 64)):
        cell = np.diag(rng.uniform(1.0, 4.0, 3)) + rng.uniform(-0.4, 0.4, (3, 3))
        dL = cell / np.array(n, dtype=np.float64)[None, :]
        off = rng.uniform(-0.5, 0.5, 3)
        ext = np.abs(cell).sum(axis=1)
        for _ in range(12):
            llc = off + rng.uniform(-0.6, 0.9, 3) * ext
            bc = np.diag(rng.uniform(-0.3, 0.7, 3) * ext) + rng.uniform(-0.05, 0.05, (3, 3))
            assert np.array_equal(ctx.box_place(n, dL, off, llc, bc), orc.box_place(n, dL, off, llc, bc)), (n, llc, bc)


@pytest.mark.gpu
def test_host_hierarchy_on_skewed_cell_and_forced_device(monkeypatch):
    g = load("box_place")
    n, cell, dL, off = _case(g, "skew")
    got = _host_places(n, cell, off, g["skew_llc"], g["skew_box_cell"])      # skewed: grid_add_box asks the device
    assert np.array_equal(got, g["skew_place"])
    n, cell, dL, off = _case(g, "ortho")
    monkeypatch.setattr(G, "BOX_PLACEMENT", "device")
    got = _host_places(n, cell, off, g["ortho_llc"], g["ortho_box_cell"])
    assert np.array_equal(got, g["ortho_place"])


@pytest.mark.gpu
def test_device_full_size_level():
    """config-3 size: every cell of 512^3 is visited; the plates and the block of the benchmark input"""
    from multigrid_amd import lib
    ctx = lib.shared_context()
    n = (512, 512, 512)
    L = 51.2
    dL = np.diag([L / 512] * 3)
    off = np.zeros(3)
    for llc, ext in (((0.25 * L, 0.25 * L, 0.0), (0.5 * L, 0.5 * L, 0.05 * L)),
                     ((0.4 * L, 0.4 * L, 0.05 * L), (0.2 * L, 0.2 * L, 0.9 * L)),
                     ((-1.0, -1.0, -1.0), (2 * L, 2 * L, 2 * L))):
        top = G.MgGrid(np.float64)
        G.init_grid(top, n, np.diag([L] * 3), 1, tol=1e-4)
        G.grid_add_box(top, np.array(llc), np.diag(ext), 1.0, 1.0, True)          # per-axis host arithmetic
        assert np.array_equal(ctx.box_place(n, dL, off, np.array(llc), np.diag(ext)), top.box[0].place)
