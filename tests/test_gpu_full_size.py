"""GPU, at the size the benchmark is quoted on (512^3 fp64, mixed periodic / Neumann / Dirichlet, two plate electrodes and a
rho = 3 block): set-up, red-black sweeps with both reductions, restriction to 256^3 and prolongation back, each against
the oracle on the same data, bit for bit; the two sweep kernels against each other; two z slabs against the whole level."""
import os
import sys

import numpy as np
import pytest
import torch          # noqa: F401  before the HIP library: one HIP runtime per process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc                                  # noqa: E402
from tests.helpers import bits_equal, oracle_levels               # noqa: E402

pytestmark = pytest.mark.gpu

N = 512


@pytest.fixture(scope="module")
def hierarchy():
    from bench import config3_input
    from multigrid_amd import mgio
    return mgio.iomg_read(mgio.MgIO(text=config3_input((N, N, N))), dtype=np.float64)


@pytest.fixture(scope="module")
def ctx():
    from multigrid_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def _device(ctx, g, z0=0, n3=None):
    from multigrid_amd import grid as G
    lv = ctx.level(g.layer, [int(v) for v in g.n], g.dtype, z0, n3)
    lv.set_params(g.sor, [float(v) for v in g.a], g.bc6())
    lv.set_boxes(*G.box_table(g))
    return lv


def test_sweeps_and_transfers_match_oracle(ctx, hierarchy):
    from multigrid_amd import lib
    top = hierarchy
    child = top.child
    assert [int(v) for v in top.n] == [N] * 3 and [int(v) for v in child.n] == [N // 2] * 3
    Lp, Lc = oracle_levels(top)[:2]
    V = np.zeros(Lp.shape(), dtype=np.float64)
    orc.setup(Lp, V, True)
    dp, dc = _device(ctx, top), _device(ctx, child)
    try:
        dp.bring_back()
        dp.setup(True)
        assert bits_equal(dp.download(), V)
        assert dp.count_nonconst() == orc.count_nonconst(Lp)
        for _ in range(2):
            e2 = orc.sweep(Lp, V, orc.ORDER_RB)
            s, units = orc.abs_sum_exact(Lp, V)
            dp.sweep(1, lib.RED_ERR | lib.RED_SUM)
            assert dp.result() == (e2, s, units)
        assert bits_equal(dp.download(), V)
        # restriction (grid_restriction_full + set-up of the child) and prolongation back
        Vc = np.zeros(Lc.shape(), dtype=np.float64)
        orc.restrict(Lp, V, Lc, Vc)
        dc.bring_back()
        lib.restrict(dp, dc, top.RES_method)
        assert bits_equal(dc.download(), Vc)
        orc.sweep(Lc, Vc, orc.ORDER_RB)
        dc.sweep(1, 0)
        Vp = np.zeros(Lp.shape(), dtype=np.float64)
        orc.prolong(Lc, Vc, Lp, Vp)
        dp.bring_back()
        lib.prolong(dc, dp, child.PRO_method)
        assert bits_equal(dp.download(), Vp)
    finally:
        dp.close()
        dc.close()


def test_two_kernels_and_two_slabs_agree(ctx, hierarchy):
    """eight sweeps from the set-up state: fused kernel, two-pass kernel and a pair of z slabs (halo planes copied on
    the device; every other sweep in two launches, edge pieces first) must hold the same bits and report the same
    reductions"""
    from multigrid_amd import lib, slab
    top = hierarchy
    whole = _device(ctx, top)
    parts = [(0, 300), (300, 212)]
    slabs = [_device(ctx, top, z0, k) for z0, k in parts]
    try:
        for lv in [whole] + slabs:
            lv.bring_back()
            lv.setup(True)

        def exchange():
            ctx.sync()
            lo, up = slabs
            slab.device_plane_tensor(up, -1, 2).copy_(slab.device_plane_tensor(lo, lo.n3_local - 1, 2))
            slab.device_plane_tensor(lo, lo.n3_local + 1, 2).copy_(slab.device_plane_tensor(up, 1, 2))
            torch.cuda.synchronize()
            for lv in slabs:
                lv.halo_written()

        exchange()
        res = {}
        red = lib.RED_ERR | lib.RED_SUM
        for it in range(8):
            whole.sweep(1, red)
            res[it] = whole.result()
            parts_res = []
            for lv in slabs:
                if it % 2:        # as under a halo exchange: the pieces that hold the edge planes first, then the rest
                    lv.sweep_part(lib.PART_EDGE, red)
                    lv.sweep_part(lib.PART_INNER, red)
                else:
                    lv.sweep(1, red)
                parts_res.append(lv.result())
            exchange()
            assert max(r[0] for r in parts_res) == res[it][0]
            assert sum(r[2] for r in parts_res) == res[it][2]
        W = whole.download()
        for lv, (z0, k) in zip(slabs, parts):
            assert bits_equal(lv.download()[1:-1], W[z0 + 1:z0 + k + 1])
        for lv in slabs:
            lv.close()
        slabs = []
        # the same eight sweeps with the two-pass kernel
        ctx.set_kernel(lib.KERNEL_TWO_PASS)
        other = _device(ctx, top)
        slabs = [other]
        other.bring_back()
        other.setup(True)
        for it in range(8):
            other.sweep(1, red)
            assert other.result() == res[it]
        assert bits_equal(other.download(), W)
    finally:
        ctx.set_kernel(0)
        whole.close()
        for lv in slabs:
            lv.close()


def test_full_extent_slabs_2048x2048x256(ctx):
    """two z slabs of 2048 x 2048 x 256 cells (the slab an 8-GPU run of the 2048^3 periodic configuration gives every
    rank; here of a 2048 x 2048 x 512 level so that the undivided level fits beside them), fused kernel, one slab in one
    launch and one in edge + inner launches, against the undivided level swept by the two-pass kernel: same bits after
    two sweeps, same reductions.  Exercises the in-plane offsets and tile lists at that extent."""
    from bench import config5_input
    from multigrid_amd import lib, mgio, slab
    top = mgio.iomg_read(mgio.MgIO(text=config5_input((2048, 2048, 512))), dtype=np.float64)
    assert [int(v) for v in top.n] == [2048, 2048, 512]
    parts = [(0, 256), (256, 256)]
    red = lib.RED_ERR | lib.RED_SUM
    made = []
    try:
        ctx.set_kernel(lib.KERNEL_TWO_PASS)
        whole = _device(ctx, top)
        made.append(whole)
        whole.bring_back()
        whole.setup(True)
        res = []
        for _ in range(2):
            whole.sweep(1, red)
            res.append(whole.result())
        ctx.set_kernel(0)
        slabs = [_device(ctx, top, z0, k) for z0, k in parts]
        made += slabs
        lo, up = slabs
        for lv in slabs:
            lv.bring_back()
            lv.setup(True)
        ctx.sync()
        # periodic C faces of the set-up (whole planes), then the halo planes of both sides
        slab.device_plane_tensor(lo, 0, 1).copy_(slab.device_plane_tensor(up, up.n3_local, 1))
        slab.device_plane_tensor(up, up.n3_local + 1, 1).copy_(slab.device_plane_tensor(lo, 1, 1))

        def exchange():
            ctx.sync()
            slab.device_plane_tensor(up, -1, 2).copy_(slab.device_plane_tensor(lo, lo.n3_local - 1, 2))
            slab.device_plane_tensor(lo, lo.n3_local + 1, 2).copy_(slab.device_plane_tensor(up, 1, 2))
            torch.cuda.synchronize()

        exchange()
        for lv in slabs:
            lv.halo_written()
        for it in range(2):
            lo.sweep(1, red)
            up.sweep_part(lib.PART_EDGE, red)
            up.sweep_part(lib.PART_INNER, red)
            rs = [lv.result() for lv in slabs]
            exchange()
            assert max(r[0] for r in rs) == res[it][0]
            assert sum(r[2] for r in rs) == res[it][2]
        for lv, (z0, k) in zip(slabs, parts):
            # rows of a plane are pitched: compare the ghosted extent, not the padding behind it
            mine = slab.device_plane_tensor(lv, 1, k).view(torch.int64).view(k, 2050, -1)[:, :, :2050]
            ref = slab.device_plane_tensor(whole, z0 + 1, k).view(torch.int64).view(k, 2050, -1)[:, :, :2050]
            assert bool(torch.equal(mine, ref))
        assert res[1][1] > 0.0
    finally:
        ctx.set_kernel(0)
        for lv in made:
            lv.close()


def _random_state(L, seed):
    """a field of noise (every cell non-zero: a wrong neighbour, weight or plane shows in the bits), boxes imprinted and
    ghost faces filled as grid_setup leaves them"""
    rng = np.random.default_rng(seed)
    V = rng.standard_normal(L.shape()).astype(np.float64)
    orc.setup(L, V, False)
    return V


def _pair_passes_against_oracle(ctx, g, L, seed):
    """`mgx_level_sweep(lv, 2)` and `(lv, 4)` under the automatic kernel selection (the two-sweeps-per-pass kernel with
    the tile lists the library builds for this size) against as many oracle sweeps: field, max dV^2, exact sum"""
    from multigrid_amd import lib
    V = _random_state(L, seed)
    lv = _device(ctx, g)
    try:
        lv.upload(V)
        red = lib.RED_ERR | lib.RED_SUM
        for nsw in (2, 4):
            for _ in range(nsw):
                e2 = orc.sweep(L, V, orc.ORDER_RB)
            s, units = orc.abs_sum_exact(L, V)
            lv.sweep(nsw, red)
            assert lv.result() == (e2, s, units), nsw
            assert bits_equal(lv.download(), V), nsw
    finally:
        lv.close()


def test_pair_pass_as_selected_at_512_matches_oracle(ctx, hierarchy):
    """the kernel behind gs_V's legs (m_gs_CDS.F90:240-242, 271-273) at the size and with the tile lists (whole chunks,
    halves, quarters, pieces that boxes reach) the V-cycle benchmark runs it with"""
    top = hierarchy
    ctx.set_kernel(0)
    _pair_passes_against_oracle(ctx, top, oracle_levels(top)[0], 512)


def test_pair_pass_cut_tail_2048x2048x64_matches_oracle(ctx):
    """one chunk of 64 planes per tile column, 1504 columns: the lists with halves and quarters of chunks on a level
    of one chunk; rows of 2048 cells (16 tile columns)"""
    from bench import config5_input
    from multigrid_amd import mgio
    top = mgio.iomg_read(mgio.MgIO(text=config5_input((2048, 2048, 64))), dtype=np.float64)
    assert [int(v) for v in top.n] == [2048, 2048, 64]
    ctx.set_kernel(0)
    _pair_passes_against_oracle(ctx, top, oracle_levels(top)[0], 2048)
