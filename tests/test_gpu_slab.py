"""GPU: z-slab levels (mgx_level_create_slab + one fused sweep per call + halo planes copied by the caller) against
the undivided level on the same GPU and against the oracle, bit for bit.  Slabs start at odd and even z."""
import os
import sys

import numpy as np
import pytest
import torch          # noqa: F401  before the HIP library: the process must end up with one HIP runtime, torch's

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _levels(ctx, n, dtype, parts, bc, boxes, sor=1.6):
    from multigrid_amd import lib
    out = []
    for z0, k in parts:
        lv = ctx.level(1, n, dtype, z0, k)
        lv.set_params(sor, (0.3, 0.3, 0.4), bc)
        lv.set_boxes(*boxes)
        out.append(lv)
    return out


# the last case is tall enough for several z chunks per slab, so that the inner launch has tiles of its own
@pytest.mark.parametrize("two_launches", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("parts", [[(0, 13), (13, 11)], [(0, 8), (8, 7), (15, 9)], [(0, 135), (135, 105)]])
def test_slab_sweeps_equal_undivided(dtype, parts, two_launches):
    import torch
    from multigrid_amd import lib, slab
    from oracle import oracle as orc
    n = (150, 45, sum(k for _, k in parts))
    bc = (1, 1, 1, 1, 3, 2)
    place = np.array([[40, 110, 10, 30, 1, 2], [40, 110, 10, 30, n[2] - 1, n[2]], [60, 90, 15, 25, 3, n[2] - 2]], dtype=np.int32)
    boxes = (place, [0.5, -0.5, 0.0], [1.0, 1.0, 3.0], [1, 1, 0])
    ctx = lib.Context(0, None)
    whole = _levels(ctx, n, dtype, [(0, n[2])], bc, boxes)[0]
    slabs = _levels(ctx, n, dtype, parts, bc, boxes)
    rng = np.random.default_rng(5)
    V0 = rng.standard_normal((n[2] + 2, n[1] + 2, n[0] + 2)).astype(dtype)
    whole.upload(V0)
    whole.setup(False)
    L = orc.Level(n=n, bc=bc, sor=1.6, a=(0.3, 0.3, 0.4),
                  boxes=[orc.Box(tuple(int(v) for v in p), v_, r_, bool(c_)) for p, v_, r_, c_ in zip(place, *boxes[1:])])
    Vref = V0.copy()
    orc.setup(L, Vref, False)
    assert np.array_equal(whole.download().view(np.uint8), Vref.view(np.uint8))

    def exchange():
        ctx.sync()
        for lo, up in zip(slabs[:-1], slabs[1:]):
            slab.device_plane_tensor(up, -1, 2).copy_(slab.device_plane_tensor(lo, lo.n3_local - 1, 2))
            slab.device_plane_tensor(lo, lo.n3_local + 1, 2).copy_(slab.device_plane_tensor(up, 1, 2))
        torch.cuda.synchronize()

    for lv, (z0, k) in zip(slabs, parts):
        lv.upload(np.ascontiguousarray(V0[z0:z0 + k + 2]))
        lv.setup(False)
    exchange()
    for lv in slabs:
        lv.halo_written()
    for it in range(4):
        e_ref = orc.sweep(L, Vref, orc.ORDER_RB)
        s_ref, u_ref = orc.abs_sum_exact(L, Vref)
        whole.sweep(1, 3)
        e_w, s_w, u_w = whole.result()
        assert (e_w, u_w) == (e_ref, u_ref)
        if two_launches:     # edge chunks first, halo planes copied, then the rest: what a multi-GPU host overlaps
            for lv in slabs:
                lv.sweep_part(lib.PART_EDGE, 3)
            exchange()
            for lv in slabs:
                lv.sweep_part(lib.PART_INNER, 3)
        else:
            for lv in slabs:
                lv.sweep(1, 3)
            exchange()
        res = [lv.result() for lv in slabs]
        assert max(r[0] for r in res) == e_ref
        assert sum(r[2] for r in res) == u_ref
        for lv, (z0, k) in zip(slabs, parts):
            got = lv.download()
            assert np.array_equal(got.view(np.uint8), Vref[z0:z0 + k + 2].view(np.uint8)), f"sweep {it} slab z0={z0}"
    ctx.close()


# two sweeps in one pass on every slab: four halo planes per side, copied once per pass
@pytest.mark.parametrize("two_launches", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("parts", [[(0, 13), (13, 11)], [(0, 8), (8, 7), (15, 9)], [(0, 135), (135, 105)],
                                   [(0, 41), (41, 64), (105, 35)]])
def test_slab_pair_passes_equal_undivided(dtype, parts, two_launches):
    _pair_passes(dtype, parts, two_launches, (150, 45), (1, 1, 1, 1, 3, 2))


# the coarse levels of a decomposed hierarchy: slabs of four or five planes, a handful of cells across
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("parts,n12,bc", [([(0, 5), (5, 5)], (8, 6), (2, 2, 2, 2, 2, 2)), ([(0, 4), (4, 4), (8, 6)], (16, 12), (2, 2, 1, 1, 1, 1)),
                                          ([(0, 10), (10, 10)], (16, 12), (1, 1, 2, 2, 3, 3)), ([(0, 20), (20, 20)], (32, 24), (2, 2, 2, 2, 1, 1))])
def test_thin_slab_pair_passes_equal_undivided(dtype, parts, n12, bc):
    _pair_passes(dtype, parts, True, n12, bc)


def _pair_passes(dtype, parts, two_launches, n12, bc):
    import torch
    from multigrid_amd import lib, slab
    from oracle import oracle as orc
    n = (n12[0], n12[1], sum(k for _, k in parts))
    x0, x1, y0, y1 = max(1, n[0] * 4 // 15), max(2, n[0] * 11 // 15), max(1, n[1] * 2 // 9), max(2, n[1] * 2 // 3)
    place = np.array([[x0, x1, y0, y1, 1, 2], [x0, x1, y0, y1, n[2] - 1, n[2]],
                      [(x0 + x1) // 2, x1 - 1, (y0 + y1) // 2, y1 - 1, 3, n[2] - 2]], dtype=np.int32)
    boxes = (place, [0.5, -0.5, 0.0], [1.0, 1.0, 3.0], [1, 1, 0])
    ctx = lib.Context(0, None)
    pair_ctx = lib.Context(0, None)
    pair_ctx.set_kernel(lib.KERNEL_PAIR)           # the undivided level takes the pair pass at this size too
    whole = _levels(pair_ctx, n, dtype, [(0, n[2])], bc, boxes)[0]
    slabs = _levels(ctx, n, dtype, parts, bc, boxes)
    rng = np.random.default_rng(8)
    V0 = rng.standard_normal((n[2] + 2, n[1] + 2, n[0] + 2)).astype(dtype)
    whole.upload(V0)
    whole.setup(False)
    L = orc.Level(n=n, bc=bc, sor=1.6, a=(0.3, 0.3, 0.4),
                  boxes=[orc.Box(tuple(int(v) for v in p), v_, r_, bool(c_)) for p, v_, r_, c_ in zip(place, *boxes[1:])])
    Vref = V0.copy()
    orc.setup(L, Vref, False)

    def exchange():
        ctx.sync()
        for lo, up in zip(slabs[:-1], slabs[1:]):
            slab.device_plane_tensor(up, -3, 4).copy_(slab.device_plane_tensor(lo, lo.n3_local - 3, 4))
            slab.device_plane_tensor(lo, lo.n3_local + 1, 4).copy_(slab.device_plane_tensor(up, 1, 4))
        torch.cuda.synchronize()

    for lv, (z0, k) in zip(slabs, parts):
        lv.upload(np.ascontiguousarray(V0[z0:z0 + k + 2]))
        lv.setup(False)
    ctx.sync()
    first, last = slabs[0], slabs[-1]        # periodic z: the ghost planes of the set-up that only the far slab can supply
    if bc[4] == lib.BC_PERIODIC:
        slab.device_plane_tensor(first, 0, 1).copy_(slab.device_plane_tensor(last, last.n3_local, 1))
    if bc[5] == lib.BC_PERIODIC:
        slab.device_plane_tensor(last, last.n3_local + 1, 1).copy_(slab.device_plane_tensor(first, 1, 1))
    exchange()
    for lv in slabs:
        lv.halo_written()
    for it in range(3):
        orc.sweep(L, Vref, orc.ORDER_RB)
        e_ref = orc.sweep(L, Vref, orc.ORDER_RB)
        s_ref, u_ref = orc.abs_sum_exact(L, Vref)
        whole.sweep(2, 3)
        e_w, s_w, u_w = whole.result()
        assert (e_w, u_w) == (e_ref, u_ref)
        if two_launches:     # tiles at the slab's ends first, the four planes a side copied, then the rest
            for lv in slabs:
                lv.sweep_part(lib.PART_EDGE2, 3)
            exchange()
            for lv in slabs:
                lv.sweep_part(lib.PART_INNER2, 3)
        else:
            for lv in slabs:
                lv.sweep(2, 3)
            exchange()
        res = [lv.result() for lv in slabs]
        assert max(r[0] for r in res) == e_ref
        assert sum(r[2] for r in res) == u_ref
        for lv, (z0, k) in zip(slabs, parts):
            got = lv.download()
            assert np.array_equal(got.view(np.uint8), Vref[z0:z0 + k + 2].view(np.uint8)), f"pass {it} slab z0={z0}"
        # a single sweep in between (odd counts, grid_solve's loop) goes on from the same four planes
        e_ref = orc.sweep(L, Vref, orc.ORDER_RB)
        whole.sweep(1, 1)
        for lv in slabs:
            lv.sweep(1, 1)
        exchange()
        assert max(lv.result()[0] for lv in slabs) == e_ref == whole.result()[0]
    for lv, (z0, k) in zip(slabs, parts):
        assert np.array_equal(lv.download().view(np.uint8), Vref[z0:z0 + k + 2].view(np.uint8))
    ctx.close()
    pair_ctx.close()


def test_slab_two_pass_kernel_refused():
    from multigrid_amd import lib
    ctx = lib.Context(0, None)
    ctx.set_kernel(lib.KERNEL_TWO_PASS)
    lv = _levels(ctx, (16, 16, 16), np.float64, [(0, 8)], (2,) * 6, (np.zeros((0, 6), np.int32), [], [], []))[0]
    lv.setup(True)
    with pytest.raises(lib.MgxError):
        lv.sweep(1, 0)
    ctx.close()


@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_slab_transfers_equal_undivided(dtype, method):
    """restriction and prolongation between z slabs (aligned: every parent slab is twice its child slab) against the
    undivided levels: the halo planes the stencils reach are the neighbours' boundary planes, copied beforehand"""
    import torch
    from multigrid_amd import lib, slab
    nP, nC = (64, 40, 48), (32, 20, 24)
    bc = (1, 1, 3, 2, 2, 3)
    partsP, partsC = [(0, 20), (20, 28)], [(0, 10), (10, 14)]
    none = (np.zeros((0, 6), np.int32), [], [], [])
    boxP = (np.array([[10, 40, 5, 30, 1, 3]], dtype=np.int32), [0.5], [1.0], [1])
    boxC = (np.array([[5, 20, 3, 15, 1, 2]], dtype=np.int32), [0.5], [1.0], [1])
    ctx = lib.Context(0, None)

    def mk(n, parts, layer, boxes):
        out = []
        for z0, k in parts:
            lv = ctx.level(layer, n, dtype, z0, k)
            lv.set_params(1.5, (1 / 3,) * 3, bc)
            lv.set_boxes(*boxes)
            out.append(lv)
        return out

    P, C_ = mk(nP, [(0, nP[2])], 1, boxP)[0], mk(nC, [(0, nC[2])], 2, boxC)[0]
    Ps, Cs = mk(nP, partsP, 1, boxP), mk(nC, partsC, 2, boxC)

    def exchange(slabs):
        ctx.sync()
        for lo, up in zip(slabs[:-1], slabs[1:]):
            slab.device_plane_tensor(up, -1, 2).copy_(slab.device_plane_tensor(lo, lo.n3_local - 1, 2))
            slab.device_plane_tensor(lo, lo.n3_local + 1, 2).copy_(slab.device_plane_tensor(up, 1, 2))
        torch.cuda.synchronize()
        for lv in slabs:
            lv.halo_written()

    rng = np.random.default_rng(3)
    V0 = rng.standard_normal((nP[2] + 2, nP[1] + 2, nP[0] + 2)).astype(dtype)
    P.upload(V0)
    P.setup(False)
    for lv, (z0, k) in zip(Ps, partsP):
        lv.upload(np.ascontiguousarray(V0[z0:z0 + k + 2]))
        lv.setup(False)
    exchange(Ps)
    # restriction
    C_.bring_back()
    lib.restrict(P, C_, method)
    Wc = C_.download()
    for lv, pl in zip(Cs, Ps):
        lv.bring_back()
        lib.restrict(pl, lv, method)
    exchange(Cs)
    for lv, (z0, k) in zip(Cs, partsC):
        got = lv.download()
        assert np.array_equal(got.view(np.uint8), Wc[z0:z0 + k + 2].view(np.uint8)), f"restriction, child slab z0={z0}"
    # prolongation back (the parent is overwritten)
    lib.prolong(C_, P, method)
    Wp = P.download()
    for lv, cl in zip(Ps, Cs):
        lib.prolong(cl, lv, method)
    exchange(Ps)
    for lv, (z0, k) in zip(Ps, partsP):
        got = lv.download()
        assert np.array_equal(got.view(np.uint8), Wp[z0:z0 + k + 2].view(np.uint8)), f"prolongation, parent slab z0={z0}"
    ctx.close()
