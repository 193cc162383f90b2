"""GPU: shapes and box tables the golden vectors do not reach — levels of one to a few cells, levels spanning several
tiles in x and y and several z chunks, overlapping boxes, several boxes with rho != 1 (cells that see two of them take
the six-division path, a third one has no weight table), constant boxes inside rho boxes, more boxes than the fused
kernel's bit sets hold, values beyond the fast range of the exact sum — all against the oracle's red-black sweep,
bit for bit, both kernels, both precisions; and the error codes of the entry points."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from multigrid_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def random_boxes(rng, n, nbox, rhos=(1.0, 3.0, 0.25, 7.5)):
    boxes = []
    for _ in range(nbox):
        lo = [int(rng.integers(1, max(2, d))) for d in n]
        hi = [int(min(d, l + rng.integers(0, max(1, d // 2) + 1))) for d, l in zip(n, lo)]
        constant = bool(rng.integers(0, 3) == 0)
        rho = 1.0 if constant else float(rng.choice(rhos))
        boxes.append(orc.Box((lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]), float(rng.uniform(-1, 1)), rho, constant))
    return boxes


def run_case(ctx, n, dtype, boxes, bc, sor, kernel, sweeps=3, seed=0, scale=1.0, a=(0.3, 0.3, 0.4), spikes=0):
    from multigrid_amd import lib
    L = orc.Level(n=tuple(n), bc=tuple(bc), sor=sor, a=a, boxes=boxes)
    rng = np.random.default_rng(seed)
    V = (rng.standard_normal(L.shape()) * scale).astype(dtype)
    for _ in range(spikes):                 # a few cells far above the rest of the field
        V[tuple(int(rng.integers(1, d + 1)) for d in reversed(n))] = 5000.0 * float(rng.uniform(1, 8))
    orc.setup(L, V, False)
    ctx.set_kernel(kernel)
    lv = ctx.level(1, n, dtype)
    try:
        lv.set_params(sor, a, bc)
        lv.set_boxes([b.place for b in boxes], [b.val for b in boxes], [b.rho for b in boxes], [int(b.constant) for b in boxes])
        lv.upload(V)
        lv.setup(False)
        assert lv.count_nonconst() == orc.count_nonconst(L)
        for k in range(sweeps):
            red = (3, 1, 2, 0)[k % 4]
            e2 = orc.sweep(L, V, orc.ORDER_LEX if kernel == 4 else orc.ORDER_RB)   # kernel 4: the reference's own order
            s, big = orc.abs_sum_exact(L, V)
            lv.sweep(1, red)
            de2, ds, dbig = lv.result() if red else (None, None, None)
            W = lv.download()
            assert bits_equal(W, V), f"n={n} sweep {k}: max diff {np.abs(W.astype(np.float64) - V).max()}"
            if red & 1:
                assert de2 == e2
            if red & 2:
                assert dbig == big and ds == s
    finally:
        lv.close()
        ctx.set_kernel(0)


@pytest.mark.parametrize("kernel", [1, 2, 4])
@pytest.mark.parametrize("n", [(1, 1, 1), (2, 1, 3), (3, 3, 3), (5, 2, 7), (1, 9, 4), (17, 1, 1)])
def test_tiny_levels(ctx, n, kernel):
    rng = np.random.default_rng(sum(n))
    for dtype in (np.float64, np.float32):
        run_case(ctx, n, dtype, random_boxes(rng, n, 2), (1, 3, 2, 1, 3, 2), 1.5, kernel, seed=sum(n))


@pytest.mark.parametrize("kernel", [1, 2, 4])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_boxes_across_tiles(ctx, seed, dtype, kernel):
    """several tiles in x (n1 > 128) and y, several z chunks; up to nine overlapping boxes with four rho values"""
    rng = np.random.default_rng(seed)
    n = [(300, 70, 40), (140, 45, 75), (260, 23, 19)][seed - 1]
    run_case(ctx, n, dtype, random_boxes(rng, n, 3 * seed), (1, 1, 2, 3, 3, 2), 1.6, kernel, seed=seed)


@pytest.mark.parametrize("kernel", [1, 2, 4])
def test_nested_and_adjacent_rho_boxes(ctx, kernel):
    """three boxes with rho != 1 touching each other and a constant box inside one of them"""
    n = (150, 44, 36)
    boxes = [orc.Box((20, 70, 5, 30, 4, 20), 0.0, 3.0, False), orc.Box((71, 120, 5, 30, 4, 20), 0.0, 0.5, False),
             orc.Box((40, 100, 10, 40, 21, 30), 0.0, 9.0, False), orc.Box((30, 50, 10, 20, 8, 12), 0.75, 1.0, True),
             orc.Box((60, 90, 1, 44, 1, 3), -0.5, 1.0, True)]
    for dtype in (np.float64, np.float32):
        run_case(ctx, n, dtype, boxes, (2, 2, 3, 3, 1, 1), 1.7, kernel, sweeps=4, seed=9)


def test_more_boxes_than_bit_sets(ctx):
    """40 boxes: the automatic choice falls back to the two-pass kernel; forcing the fused one is an error"""
    from multigrid_amd import lib
    n = (40, 30, 20)
    rng = np.random.default_rng(40)
    boxes = random_boxes(rng, n, 40)
    run_case(ctx, n, np.float64, boxes, (1, 1, 2, 2, 3, 3), 1.5, 0, seed=40)
    with pytest.raises(lib.MgxError):
        run_case(ctx, n, np.float64, boxes, (1, 1, 2, 2, 3, 3), 1.5, 2, seed=40)
    run_case(ctx, n, np.float64, boxes, (1, 1, 2, 2, 3, 3), 1.5, 4, seed=40)      # the reference's order scans the table


@pytest.mark.parametrize("tiled", ["1", "0"])
def test_reference_order_by_tiles_and_by_cells(ctx, tiled, monkeypatch):
    """the two forms of the reference-order sweep (tile hyperplanes, cell hyperplanes) against the oracle's
    lexicographic sweep on a level with partial tiles in every direction"""
    from multigrid_amd import lib
    monkeypatch.setenv("MGX_LEX_TILED", tiled)
    c = lib.Context(0)                      # the switch is read when a context is created
    try:
        rng = np.random.default_rng(11)
        n = (75, 19, 21)
        run_case(c, n, np.float64, random_boxes(rng, n, 6), (1, 1, 3, 2, 2, 3), 1.8, 4, sweeps=4, seed=11)
        run_case(c, n, np.float32, [], (1, 1, 1, 1, 1, 1), 1.9, 4, sweeps=4, seed=12)
    finally:
        c.close()


@pytest.mark.parametrize("kernel", [1, 2])
def test_sum_beyond_the_fast_range(ctx, kernel):
    """|V| up to ~1e5: the two-double partial sums hand over to the integer path, same exact sum"""
    n = (140, 24, 10)
    run_case(ctx, n, np.float64, [], (1, 1, 1, 1, 1, 1), 1.2, kernel, seed=3, scale=3.0e4)


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_sum_of_tiny_values_beside_large_ones(ctx, kernel):
    """a field of values around 1e-9 (bits far below the 2^-72 grid of the sum) with a few cells beyond the fast
    range: the lanes that share a wave with a large value take the integer path and must round, not truncate"""
    n = (140, 24, 10)
    run_case(ctx, n, np.float64, [], (1, 1, 1, 1, 1, 1), 1.0, kernel, sweeps=5, seed=5, scale=1.0e-9, spikes=6)


def test_error_codes(ctx):
    from multigrid_amd import lib
    with pytest.raises(lib.MgxError) as e:
        ctx.level(1, (0, 4, 4), np.float64)
    assert e.value.code == -1
    lv = ctx.level(1, (8, 8, 8), np.float64)
    try:
        with pytest.raises(lib.MgxError) as e:
            lv.sweep(1, 0)                      # no parameters yet
        assert e.value.code == -4
        lv.set_params(1.5, (1 / 3,) * 3, (2,) * 6)
        with pytest.raises(lib.MgxError) as e:
            lv.set_boxes([(1, 2, 1, 2, 1, 2)], [0.0], [-1.0], [0])     # rho < 0 (t_mg.f90:352)
        assert e.value.code == -1
        with pytest.raises(lib.MgxError) as e:
            lv.set_boxes([(1, 2, 1, 2, 1, 2)], [2.0 ** 19], [1.0], [1])   # beyond the range of the exact sum of |V|
        assert e.value.code == -5 and "2^19" in str(e.value)
        with pytest.raises(lib.MgxError):
            lv.half_sweep(2, 1, 8, 0)
        with pytest.raises(lib.MgxError):
            lv.sweep_part(lib.PART_INNER, 0)    # inner part without its edge part
        lv.set_boxes([], [], [], [])
        lv.bring_back()
        lv.setup(True)
        lv.sweep_part(lib.PART_EDGE, 3)
        with pytest.raises(lib.MgxError) as e:
            lv.sweep_part(lib.PART_INNER, 1)    # the two parts of a sweep share one fold of the reductions
        assert e.value.code == -1
        lv.sweep_part(lib.PART_INNER, 3)
        lv.result()
        with pytest.raises(lib.MgxError):
            lv.plane_ptr(11, 1)
    finally:
        lv.close()


@pytest.mark.parametrize("spikes", [0, 5])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_sum_of_values_halfway_between_grid_points(ctx, kernel, dtype, spikes):
    """every |V| is an odd multiple of 2^-73, exactly halfway between two points of the 2^-72 grid the sum rounds to:
    all forms of the add (two-double partial sums, integer path beside values beyond the fast range, the plain
    grid_sum kernel) must break the tie to even like the oracle's integer arithmetic.  sor = 0 keeps the field as it
    is (V = 1*V + 0*avg), so the sweeps' own reductions see exactly these values."""
    from multigrid_amd import lib
    n = (140, 30, 9)
    L = orc.Level(n=n, bc=(1,) * 6, sor=0.0, a=(0.3, 0.3, 0.4), boxes=[])
    rng = np.random.default_rng(17 + spikes)
    m = rng.integers(0, 1 << 20, size=L.shape())
    V = (np.ldexp((2 * m + 1).astype(np.float64), -73) * rng.choice([-1.0, 1.0], size=L.shape())).astype(dtype)
    assert np.array_equal(V.astype(np.float64) * 2.0 ** 73 % 2, np.ones(L.shape()))      # nothing was rounded away
    for _ in range(spikes):
        V[tuple(int(rng.integers(1, d + 1)) for d in reversed(n))] = 4096.0
    s, big = orc.abs_sum_exact(L, V)
    ctx.set_kernel(kernel)
    lv = ctx.level(1, n, dtype)
    try:
        lv.set_params(0.0, (0.3, 0.3, 0.4), (1,) * 6)
        lv.set_boxes([], [], [], [])
        lv.upload(V)
        assert lv.abs_sum_limbs() == big
        for nsw in (1, 2):
            lv.sweep(nsw, lib.RED_ERR | lib.RED_SUM)
            de2, ds, dbig = lv.result()
            assert bits_equal(lv.download(), V)
            assert de2 == 0.0 and dbig == big and ds == s
    finally:
        lv.close()
        ctx.set_kernel(0)
