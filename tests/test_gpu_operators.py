"""GPU parity of the level operators: HIP library (through the C ABI) vs the CPU oracle and the
golden vectors of the reference.  Bar: bit-exact, both precisions."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import DT, bits_equal, boxes_of, level_of, load

pytestmark = pytest.mark.gpu

SWEEPS = ["sweep_24_mixed", "sweep_33x20x17_w", "sweep_16_zero_start", "sweep_12x10x8_periodic"]
XFERS = ["xfer_32_16", "xfer_24x16x40", "xfer_odd_30x19x30", "xfer_nonhalved_32x32x64"]


@pytest.fixture(scope="module")
def ctx():
    from multigrid_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def dev_level(ctx, L, dtype, layer=1):
    lv = ctx.level(layer, L.n, dtype)
    lv.set_params(L.sor, L.a, L.bc)
    lv.set_boxes([b.place for b in L.boxes], [b.val for b in L.boxes], [b.rho for b in L.boxes],
                 [int(b.constant) for b in L.boxes])
    return lv


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", SWEEPS)
def test_reference_order_sweep_equals_the_reference(ctx, name, prec):
    """MGX_KERNEL_LEX against the fields the compiled reference left after each gs_step (no oracle in between)"""
    from multigrid_amd import lib
    g = load(f"{name}_{prec}")
    L = level_of(g)
    ctx.set_kernel(lib.KERNEL_LEX)
    try:
        lv = dev_level(ctx, L, g["V0"].dtype)
        lv.upload(g["V0"].copy())
        for k in range(int(g["k"])):
            lv.sweep(1, lib.RED_ERR | lib.RED_SUM)
            de2, ds, _ = lv.result()
            assert bits_equal(lv.download(), g["V"][k]), f"{name} {prec} sweep {k}"
            assert np.sqrt(g["V0"].dtype.type(de2)) == g["V0"].dtype.type(g["err"][k])       # grid%err = sqrt(err), m_gs_CDS.F90:313
            assert abs(ds - float(g["sum"][k])) <= (1e-12 if prec == "f64" else 2e-4) * abs(float(g["sum"][k]))
        lv.close()
    finally:
        ctx.set_kernel(0)


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", SWEEPS)
def test_rb_sweep_matches_oracle(ctx, name, prec, kernel):
    from multigrid_amd import lib
    g = load(f"{name}_{prec}")
    L = level_of(g)
    V = g["V0"].copy()
    ctx.set_kernel(kernel)
    try:
        lv = dev_level(ctx, L, V.dtype)
        lv.upload(V)
        assert lv.count_nonconst() == int(g["count"])
        assert lv.tolerance(1e-4) == float(g["tolerance"])
        assert lv.abs_sum_limbs() == orc.abs_sum_exact(L, V)[1]
        for k in range(int(g["k"])):
            e2 = orc.sweep(L, V, orc.ORDER_RB)
            s, big = orc.abs_sum_exact(L, V)
            try:
                lv.sweep(1, lib.RED_ERR | lib.RED_SUM)
            except lib.MgxError as e:
                if e.code == -5:
                    pytest.skip("kernel does not support this shape")
                raise
            de2, ds, dbig = lv.result()
            W = lv.download()
            assert bits_equal(W, V), f"{name} {prec} sweep {k}: {np.abs(W - V).max()}"
            assert de2 == e2
            assert dbig == big and ds == s
        lv.close()
    finally:
        ctx.set_kernel(0)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("method", ["full", "half"])
@pytest.mark.parametrize("name", XFERS)
def test_transfer_matches_reference(ctx, name, method, prec):
    from multigrid_amd import lib
    g = load(f"{name}_{method}_{prec}")
    m = int(g["method"])
    P = level_of(g)
    Cl = level_of(g, n_key="nc", boxes=[])
    dp, dc = dev_level(ctx, P, DT[prec], 1), dev_level(ctx, Cl, DT[prec], 2)
    dp.upload(g["res_parent"])
    dc.bring_back()
    lib.restrict(dp, dc, m)
    assert bits_equal(dc.download(), g["res_child"])
    dc.upload(g["pro_child"])
    dp.bring_back()
    lib.prolong(dc, dp, m)
    assert bits_equal(dp.download(), g["pro_parent"])
    dp.close()
    dc.close()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_setup_all_bc_combinations(ctx, prec):
    g = load(f"setup_6x5x4_{prec}")
    boxes = boxes_of(g)
    n = tuple(int(v) for v in g["n"])
    lv = ctx.level(1, n, DT[prec])
    lv.set_boxes([b.place for b in boxes], [b.val for b in boxes], [b.rho for b in boxes],
                 [int(b.constant) for b in boxes])
    for bc, want in zip(g["bc"], g["V_out"]):
        lv.set_params(1.8, (1 / 3, 1 / 3, 1 / 3), [int(v) for v in bc])
        lv.upload(g["V_in"])
        lv.setup(False)
        assert bits_equal(lv.download(), want), f"bc={bc}"
    lv.set_params(1.8, (1 / 3, 1 / 3, 1 / 3), [2] * 6)
    lv.setup(True)
    L = orc.Level(n=n, bc=(2,) * 6, boxes=boxes)
    V = np.ones(L.shape(), DT[prec])
    orc.setup(L, V, True)
    assert bits_equal(lv.download(), V)
    lv.close()


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("i", [8, 9, 10, 11, 12, 13, 14, 15, 128])
def test_reference_test1_constant_field(ctx, i, prec):
    """the reference's test/test1.f90 on the GPU: parent i^3 holding 1, child ((i + mod(i,2))/2 - 1)^3 (not a halving,
    generic index maps); restriction and prolongation against the oracle bit for bit, and the known answer: maximum 1"""
    from multigrid_amd import lib
    dt = DT[prec]
    nc = (i + i % 2) // 2 - 1
    for method in (1, 2):
        eps = 0.0 if method == 1 else 4 * float(np.finfo(dt).eps)
        P = orc.Level(n=(i, i, i), res_method=method)
        Cl = orc.Level(n=(nc, nc, nc), pro_method=method)
        dp, dc = dev_level(ctx, P, dt, 1), dev_level(ctx, Cl, dt, 2)
        try:
            Vp = np.ones(P.shape(), dtype=dt)
            Vc = np.zeros(Cl.shape(), dtype=dt)
            orc.restrict(P, Vp, Cl, Vc)
            dp.upload(Vp)
            dc.bring_back()
            lib.restrict(dp, dc, method)
            W = dc.download()
            assert bits_equal(W, Vc)
            assert abs(float(W.max()) - 1.0) <= eps
            Vc[...] = 1
            dc.upload(Vc)
            orc.prolong(Cl, Vc, P, Vp)
            dp.bring_back()
            lib.prolong(dc, dp, method)
            W = dp.download()
            assert bits_equal(W, Vp)
            assert abs(float(W[1:-1, 1:-1, 1:-1].max()) - 1.0) <= eps
        finally:
            dp.close()
            dc.close()
