"""The CPU oracle (lexicographic mode) against golden vectors produced by the reference itself.

Fixtures: tests/golden/*.npz, written by oracle/make_golden.py from the reference compiled with
amdflang (1 thread).  The bar is bit-exact for both precisions.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import DT, bits_equal, boxes_of, level_of, load

SWEEPS = ["sweep_24_mixed", "sweep_33x20x17_w", "sweep_16_zero_start", "sweep_12x10x8_periodic"]
XFERS = ["xfer_32_16", "xfer_24x16x40", "xfer_odd_30x19x30", "xfer_nonhalved_32x32x64"]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", SWEEPS)
def test_lex_sweep_matches_reference(name, prec):
    g = load(f"{name}_{prec}")
    L = level_of(g)
    V = g["V0"].copy()
    assert V.dtype == DT[prec]
    assert orc.count_nonconst(L) == int(g["count"])
    assert orc.abs_sum_seq(L, V) == float(g["sum0"])
    L.tol = 1e-4
    assert orc.tolerance(L, V.dtype) == float(g["tolerance"])
    for k in range(int(g["k"])):
        e2 = orc.sweep(L, V, orc.ORDER_LEX)
        assert bits_equal(V, g["V"][k]), f"{name} {prec} sweep {k}"
        err = float(np.sqrt(V.dtype.type(e2)))
        assert err == float(g["err"][k])
        assert orc.abs_sum_seq(L, V) == float(g["sum"][k])


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("method", ["full", "half"])
@pytest.mark.parametrize("name", XFERS)
def test_transfer_matches_reference(name, method, prec):
    g = load(f"{name}_{method}_{prec}")
    m = int(g["method"])
    P = level_of(g, res_method=m, pro_method=m)
    Cl = level_of(g, n_key="nc", boxes=[], res_method=m, pro_method=m)
    Vc = np.zeros(Cl.shape(), DT[prec])
    orc.restrict(P, g["res_parent"].copy(), Cl, Vc)
    assert bits_equal(Vc, g["res_child"])
    Vp = np.zeros(P.shape(), DT[prec])
    orc.prolong(Cl, g["pro_child"].copy(), P, Vp)
    assert bits_equal(Vp, g["pro_parent"])


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_setup_all_bc_combinations(prec):
    g = load(f"setup_6x5x4_{prec}")
    boxes = boxes_of(g)
    for bc, want in zip(g["bc"], g["V_out"]):
        L = orc.Level(n=tuple(int(v) for v in g["n"]), bc=tuple(int(v) for v in bc), boxes=boxes)
        V = g["V_in"].copy()
        orc.setup(L, V, init=False)
        assert bits_equal(V, want), f"bc={bc}"


def test_exact_sum_is_order_independent_and_close_to_seq():
    from fractions import Fraction
    rng = np.random.default_rng(7)
    L = orc.Level(n=(20, 13, 9))
    V = rng.standard_normal(L.shape()) * np.exp(rng.uniform(-60, 5, L.shape()))
    s, big = orc.abs_sum_exact(L, V)
    exact = 0
    for x in map(float, np.abs(V[1:-1, 1:-1, 1:-1]).ravel()):
        exact += round(Fraction(x) * 2 ** 72)          # python rounds halves to even
    assert big == exact
    assert s == exact / 2 ** 72                         # int / int is correctly rounded
    Vp = V.copy()
    Vp[1:-1, 1:-1, 1:-1] = Vp[1:-1, 1:-1, 1:-1][::-1, ::-1, ::-1]
    assert orc.abs_sum_exact(L, Vp)[1] == big
    assert abs(s - orc.abs_sum_seq(L, V)) <= 1e-12 * s


@pytest.mark.parametrize("i", [8, 9, 10, 11, 12, 13, 14, 15, 128])
def test_reference_test1_constant_field(i):
    """the reference's own known answer (test/test1.f90:27-75, printed with three digits): a parent of i^3 cells
    holding 1 everywhere, a child of ((i + mod(i,2))/2 - 1)^3 cells (not a halving); the maximum after restriction
    and after prolongation is 1 — exactly with the /64 weights, to rounding with the /56 weights — and no cell
    an operator wrote exceeds it"""
    nc = (i + i % 2) // 2 - 1
    for method in (1, 2):
        eps = 0.0 if method == 1 else 4 * np.finfo(np.float64).eps
        P = orc.Level(n=(i, i, i), res_method=method)
        Cl = orc.Level(n=(nc, nc, nc), pro_method=method)
        Vp = np.ones(P.shape(), dtype=np.float64)
        Vc = np.zeros(Cl.shape(), dtype=np.float64)
        orc.restrict(P, Vp, Cl, Vc)
        assert abs(Vc.max() - 1.0) <= eps
        if method == 2:                       # the half-weight loop writes every child cell
            assert np.abs(Vc[1:-1, 1:-1, 1:-1] - 1.0).max() <= eps
        Vc[...] = 1.0
        orc.prolong(Cl, Vc, P, Vp)            # child ghosts hold 1 as well: every stencil sees only ones
        assert np.abs(Vp[1:-1, 1:-1, 1:-1] - 1.0).max() <= eps
