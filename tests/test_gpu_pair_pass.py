"""GPU parity of the two-sweeps-per-pass kernel (gs_V's back-to-back sweeps, m_gs_CDS.F90:240-242, 271-273):
`mgx_level_sweep(lv, k)` with pairs of sweeps taken in one pass over the field against the CPU oracle's k
red-black sweeps, and against the one-sweep-per-pass kernel.  Bar: bit-exact fields, max dV^2 and exact sum."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import bits_equal, level_of, load

pytestmark = pytest.mark.gpu

KERNEL_FUSED, KERNEL_PAIR = 2, 3


@pytest.fixture(scope="module")
def ctx():
    from multigrid_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def dev_level(ctx, L, dtype):
    lv = ctx.level(1, L.n, dtype)
    lv.set_params(L.sor, L.a, L.bc)
    lv.set_boxes([b.place for b in L.boxes], [b.val for b in L.boxes], [b.rho for b in L.boxes],
                 [int(b.constant) for b in L.boxes])
    return lv


def random_level(n, bc, boxes, sor=1.7, a=(1 / 3,) * 3):
    return orc.Level(n=n, bc=bc, sor=sor, a=a, boxes=boxes)


def run_case(ctx, L, V0, nsweeps_list):
    from multigrid_amd import lib
    for nsw in nsweeps_list:
        V = V0.copy()
        for _ in range(nsw):
            e2 = orc.sweep(L, V, orc.ORDER_RB)
        s, big = orc.abs_sum_exact(L, V)
        out = {}
        for kernel in (KERNEL_PAIR, KERNEL_FUSED):
            ctx.set_kernel(kernel)
            try:
                lv = dev_level(ctx, L, V0.dtype)
                lv.upload(V0)
                lv.sweep(nsw, lib.RED_ERR | lib.RED_SUM)
                de2, ds, dbig = lv.result()
                W = lv.download()
                lv.close()
            finally:
                ctx.set_kernel(0)
            out[kernel] = (W, de2, ds, dbig)
            assert bits_equal(W, V), f"kernel {kernel}, {nsw} sweeps: max diff {np.abs(W - V).max()} at {np.unravel_index(np.argmax(np.abs(W - V)), W.shape)}"
            assert de2 == e2, (kernel, nsw, de2, e2)
            assert dbig == big and ds == s, (kernel, nsw)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", ["sweep_24_mixed", "sweep_33x20x17_w", "sweep_16_zero_start", "sweep_12x10x8_periodic"])
def test_pair_pass_on_golden_inputs(ctx, name, prec):
    g = load(f"{name}_{prec}")
    run_case(ctx, level_of(g), g["V0"].copy(), [2, 3, 4])


CASES = [
    # n, bc, boxes (place, val, rho, constant)
    ((140, 50, 20), (1, 1, 3, 2, 2, 3), [((30, 100, 10, 40, 3, 6), 0.5, 1.0, True), ((60, 80, 20, 30, 8, 15), 0.0, 3.0, False)]),
    ((257, 45, 9), (2, 2, 1, 1, 3, 3), [((1, 257, 1, 3, 1, 9), -0.5, 1.0, True), ((100, 160, 10, 40, 2, 8), 0.0, 2.5, False),
                                        ((120, 200, 20, 30, 4, 6), 0.25, 0.5, False)]),
    ((64, 23, 70), (3, 2, 2, 3, 1, 1), []),
    ((128, 44, 12), (1, 1, 1, 1, 1, 1), [((5, 9, 5, 9, 5, 9), 1.0, 1.0, True)]),
    ((5, 3, 4), (2, 3, 1, 1, 2, 2), [((2, 3, 1, 2, 2, 3), 0.5, 4.0, False)]),
]


@pytest.mark.parametrize("prec", [np.float32, np.float64])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_pair_pass_tiles_and_boxes(ctx, case, prec):
    n, bc, bx = CASES[case]
    boxes = [orc.Box(tuple(p), float(v), float(r), bool(c)) for p, v, r, c in bx]
    L = random_level(n, bc, boxes)
    rng = np.random.default_rng(100 + case)
    V = rng.standard_normal((n[2] + 2, n[1] + 2, n[0] + 2)).astype(prec)
    orc.setup(L, V, False)
    run_case(ctx, L, V, [2, 5])


def test_pair_pass_sum_of_tiny_values_beside_large_ones(ctx):
    """values around 1e-9 (bits far below the 2^-72 grid of the sum) with a few cells beyond the fast range of the
    partial sums: lanes that share a wave with a large value take the integer path and must round like the rest"""
    n = (140, 44, 12)
    L = random_level(n, (1, 1, 1, 1, 1, 1), [], sor=1.0)
    rng = np.random.default_rng(7)
    V = (rng.standard_normal((n[2] + 2, n[1] + 2, n[0] + 2)) * 1.0e-9).astype(np.float64)
    for _ in range(6):
        V[tuple(int(rng.integers(1, d + 1)) for d in reversed(n))] = 5000.0 * float(rng.uniform(1, 8))
    orc.setup(L, V, False)
    run_case(ctx, L, V, [2, 4])
