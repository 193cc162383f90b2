"""Randomised parity run on the GPU box: random level shapes, boundary conditions, box tables, relaxation factors,
magnitudes, kernels and sweep counts; the library's fields, max dV^2 and exact sums against the CPU oracle, bit for
bit.  Usage: python scripts/fuzz_parity.py [cases] [seed] [transfers|slabs|solves|boxes]
(`transfers`: restriction and prolongation between random pairs of levels instead of sweeps; `slabs`: sweeps of a
level cut into z slabs, halo planes copied between them, against the oracle's sweeps of the undivided level;
`solves`: whole bottom-up and V-cycle solves of random input files against the oracle's solve; `boxes`: the device's
search for the cells of a box against the oracle's).  Prints one line per 10 cases; stops at the first mismatch
with the parameters that reproduce it."""
import sys
import time

import numpy as np

from multigrid_amd import lib
from oracle import oracle as orc


STATS = {"compared": 0, "stopped": 0, "not runnable": 0}


def random_boxes(rng, n, nbox):
    boxes = []
    for _ in range(nbox):
        lo = [int(rng.integers(1, d + 1)) for d in n]
        hi = [int(min(d, l + rng.integers(0, max(1, d // 2) + 1))) for d, l in zip(n, lo)]
        constant = bool(rng.integers(0, 3) == 0)
        rho = 1.0 if constant else float(rng.choice([1.0, 3.0, 0.25, 7.5]))
        boxes.append(orc.Box((lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]), float(rng.uniform(-1, 1)), rho, constant))
    return boxes


def bits_equal(a, b):
    u = np.uint64 if a.dtype == np.float64 else np.uint32
    return a.shape == b.shape and np.array_equal(a.view(u), b.view(u))


def one_case(ctx, rng, idx):
    dims = [1, 2, 3, 5, 8, 17, 31, 64, 65, 127, 128, 129, 140, 200, 257, 300]
    n = (int(rng.choice(dims)), int(rng.choice(dims[:12])), int(rng.choice(dims[:11])))
    if rng.integers(0, 4) == 0:
        n = tuple(int(rng.integers(1, 90)) for _ in range(3))
    dtype = np.float64 if rng.integers(0, 2) else np.float32
    kernel = int(rng.choice([0, 1, 2, 3, 4]))
    bc = tuple(int(rng.integers(1, 4)) for _ in range(6))
    for ax in range(3):                                   # a periodic face comes with its partner
        if lib.BC_PERIODIC in (bc[2 * ax], bc[2 * ax + 1]):
            bc = bc[:2 * ax] + (lib.BC_PERIODIC,) * 2 + bc[2 * ax + 2:]
    sor = float(rng.choice([1.0, 1.2, 1.5, 1.7, 0.8]))
    a = rng.uniform(0.1, 1.0, 3)
    a = tuple(float(v) for v in a / a.sum())
    nbox = int(rng.choice([0, 0, 1, 2, 4, 7]))
    boxes = random_boxes(rng, n, nbox)
    scale = float(rng.choice([1.0, 1.0, 1.0e-9, 1.0e-3, 50.0, 3.0e3]))
    desc = dict(idx=idx, n=n, dtype=np.dtype(dtype).name, kernel=kernel, bc=bc, sor=sor, a=a, nbox=nbox, scale=scale)
    L = orc.Level(n=n, bc=bc, sor=sor, a=a, boxes=boxes)
    V = (rng.standard_normal(L.shape()) * scale).astype(dtype)
    if rng.integers(0, 4) == 0:
        for _ in range(3):
            V[tuple(int(rng.integers(1, d + 1)) for d in reversed(n))] = 3000.0 * float(rng.uniform(1, 5))
    orc.setup(L, V, False)
    ctx.set_kernel(kernel)
    lv = ctx.level(1, n, dtype)
    try:
        lv.set_params(sor, a, bc)
        lv.set_boxes([b.place for b in boxes], [b.val for b in boxes], [b.rho for b in boxes], [int(b.constant) for b in boxes])
        lv.upload(V)
        lv.setup(False)
        assert lv.count_nonconst() == orc.count_nonconst(L), ("count", desc)
        for k in range(3):
            nsw = int(rng.integers(1, 5))
            red = int(rng.choice([0, 1, 2, 3, 3]))
            for _ in range(nsw):
                e2 = orc.sweep(L, V, orc.ORDER_LEX if kernel == 4 else orc.ORDER_RB)
            s, big = orc.abs_sum_exact(L, V)
            try:
                lv.sweep(nsw, red)
            except lib.MgxError as e:
                if e.code == -5 and kernel in (2, 3):     # a forced kernel may refuse a shape
                    return
                raise
            W = lv.download()
            assert bits_equal(W, V), ("field", k, nsw, red, desc)
            if red:
                de2, ds, dbig = lv.result()
                if red & 1:
                    assert de2 == e2, ("err", k, nsw, red, de2, e2, desc)
                if red & 2:
                    assert dbig == big and (ds == s or (ds != ds and s != s)), ("sum", k, nsw, red, dbig, big, desc)
            assert lv.abs_sum_limbs() == big, ("abs_sum", k, desc)
    finally:
        lv.close()
        ctx.set_kernel(0)


def dev_level(ctx, L, dtype, layer):
    lv = ctx.level(layer, L.n, dtype)
    lv.set_params(L.sor, L.a, L.bc)
    lv.set_boxes([b.place for b in L.boxes], [b.val for b in L.boxes], [b.rho for b in L.boxes],
                 [int(b.constant) for b in L.boxes])
    return lv


def one_transfer(ctx, rng, idx):
    """grid_restriction and grid_prolongation (both end in grid_setup of the level they write) between a parent and a
    child whose extents are halved, halved and rounded either way, or unrelated"""
    dims = [2, 3, 4, 7, 8, 15, 16, 30, 31, 32, 33, 64, 66, 100, 128, 130, 200, 256, 258]
    n = (int(rng.choice(dims)), int(rng.choice(dims[:13])), int(rng.choice(dims[:12])))

    def child(d):
        k = int(rng.integers(0, 5))
        c = [d // 2, (d + 1) // 2, (d + d % 2) // 2 - 1, int(rng.integers(1, d + 1)), d // 2][k]
        return max(1, c)
    nc = tuple(child(d) for d in n)
    if rng.integers(0, 3) == 0:
        nc = tuple(d // 2 if d % 2 == 0 else c for d, c in zip(n, nc))     # the plain halving wherever it exists
    dtype = np.float64 if rng.integers(0, 2) else np.float32
    method = int(rng.integers(1, 3))
    bc = tuple(int(rng.integers(1, 4)) for _ in range(6))
    for ax in range(3):
        if lib.BC_PERIODIC in (bc[2 * ax], bc[2 * ax + 1]):
            bc = bc[:2 * ax] + (lib.BC_PERIODIC,) * 2 + bc[2 * ax + 2:]
    desc = dict(idx=idx, n=n, nc=nc, dtype=np.dtype(dtype).name, method=method, bc=bc)
    P = orc.Level(n=n, bc=bc, boxes=random_boxes(rng, n, int(rng.choice([0, 1, 3]))), res_method=method, pro_method=method)
    Cl = orc.Level(n=nc, bc=bc, boxes=random_boxes(rng, nc, int(rng.choice([0, 1, 3]))), res_method=method, pro_method=method)
    dp, dc = dev_level(ctx, P, dtype, 1), dev_level(ctx, Cl, dtype, 2)
    try:
        Vp = rng.standard_normal(P.shape()).astype(dtype)
        Vc = np.zeros(Cl.shape(), dtype=dtype)
        dp.upload(Vp)
        dc.bring_back()
        orc.restrict(P, Vp, Cl, Vc)
        lib.restrict(dp, dc, method)
        assert bits_equal(dc.download(), Vc), ("restrict", desc)
        Vc = rng.standard_normal(Cl.shape()).astype(dtype)
        dc.upload(Vc)
        Vp = np.zeros(P.shape(), dtype=dtype)
        dp.bring_back()
        orc.prolong(Cl, Vc, P, Vp)
        lib.prolong(dc, dp, method)
        assert bits_equal(dp.download(), Vp), ("prolong", desc)
    finally:
        dp.close()
        dc.close()


def one_slab_case(ctx, rng, idx):
    """a level cut into two to four z slabs on one GPU (what the ranks of a multi-GPU run hold): every slab swept in
    one launch or in edge + inner launches, two halo planes per side copied between the slabs after every sweep; where
    every slab has four planes, half of the cases keep four halo planes current and take passes of two sweeps (one
    launch or edge + inner launches) with single sweeps in between; fields and reductions against the oracle's sweeps
    of the undivided level"""
    import torch
    from multigrid_amd import slab
    dims = [3, 8, 17, 64, 65, 129, 140, 257]
    n = (int(rng.choice(dims)), int(rng.choice(dims[:6])), int(rng.integers(4, 70)))
    ns = int(rng.integers(2, 5))
    while n[2] < 2 * ns:
        ns -= 1
    while True:                                         # cut points anywhere, every slab at least two planes thick
        cuts = sorted(int(c) for c in rng.choice(np.arange(2, n[2] - 1), size=ns - 1, replace=False))
        edges = [0] + cuts + [n[2]]
        if min(b - a for a, b in zip(edges[:-1], edges[1:])) >= 2:
            break
    parts = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]
    dtype = np.float64 if rng.integers(0, 2) else np.float32
    kernel = int(rng.choice([0, 2]))
    bc = tuple(int(rng.integers(2, 4)) for _ in range(6))
    sor = float(rng.choice([1.0, 1.5, 1.7]))
    a = (0.3, 0.3, 0.4)
    boxes = random_boxes(rng, n, int(rng.choice([0, 1, 3, 6])))
    desc = dict(idx=idx, n=n, parts=parts, dtype=np.dtype(dtype).name, kernel=kernel, bc=bc, sor=sor, nbox=len(boxes))
    L = orc.Level(n=n, bc=bc, sor=sor, a=a, boxes=boxes)
    V = rng.standard_normal(L.shape()).astype(dtype)
    orc.setup(L, V, False)
    ctx.set_kernel(kernel)
    slabs = []
    try:
        for z0, k in parts:
            lv = ctx.level(1, n, dtype, z0, k)
            lv.set_params(sor, a, bc)
            lv.set_boxes([b.place for b in boxes], [b.val for b in boxes], [b.rho for b in boxes], [int(b.constant) for b in boxes])
            lv.upload(V[z0:z0 + k + 2])
            slabs.append(lv)

        depth = 4 if (kernel == 0 and min(k for _, k in parts) >= 4 and rng.integers(0, 2)) else 2
        desc["depth"] = depth

        def exchange():
            ctx.sync()
            for lo, up in zip(slabs[:-1], slabs[1:]):
                slab.device_plane_tensor(up, 1 - depth, depth).copy_(slab.device_plane_tensor(lo, lo.n3_local + 1 - depth, depth))
                slab.device_plane_tensor(lo, lo.n3_local + 1, depth).copy_(slab.device_plane_tensor(up, 1, depth))
            torch.cuda.synchronize()
            for lv in slabs:
                lv.halo_written()

        exchange()
        for it in range(4):
            red = int(rng.choice([0, 1, 3, 3]))
            two = depth == 4 and rng.integers(0, 3) > 0              # a pass of two sweeps, or a single sweep in between
            if two:
                orc.sweep(L, V, orc.ORDER_RB)
            e2 = orc.sweep(L, V, orc.ORDER_RB)
            s, big = orc.abs_sum_exact(L, V)
            rs = []
            for lv in slabs:
                if rng.integers(0, 2) and not (depth == 4 and not two):
                    lv.sweep_part(lib.PART_EDGE2 if two else lib.PART_EDGE, red)
                    lv.sweep_part(lib.PART_INNER2 if two else lib.PART_INNER, red)
                else:
                    lv.sweep(2 if two else 1, red)
                if red:
                    rs.append(lv.result())
            exchange()
            if red & 1:
                assert max(r[0] for r in rs) == e2, ("err", it, desc)
            if red & 2:
                assert sum(r[2] for r in rs) == big, ("sum", it, desc)
            for lv, (z0, k) in zip(slabs, parts):
                assert bits_equal(np.ascontiguousarray(lv.download()[1:-1]), np.ascontiguousarray(V[z0 + 1:z0 + k + 1])), ("field", it, z0, desc)
    finally:
        for lv in slabs:
            lv.close()
        ctx.set_kernel(0)


def random_input(rng):
    """an input file in the reference's format (m_mg_io.F90): a box-shaped cell, one to three boxes (electrodes with a
    value, dielectric blocks with a density), random mesh, boundary conditions, cycle and transfer methods"""
    sizes = [24, 40, 41, 44, 48, 50, 64, 80, 81]      # from 40 cells on a direction is halved (t_mg.f90:183-201)
    while True:
        n = [int(rng.choice(sizes)) for _ in range(3)]
        if n[0] * n[1] * n[2] <= 160000:
            break
    cell = [d / 10.0 for d in n]
    lines = [f"cell {cell[0]} {cell[1]} {cell[2]}", f"max-layers {int(rng.integers(3, 6))}"]
    nbox = int(rng.integers(1, 4))
    lines.append(f"boxes {nbox}")
    if rng.integers(0, 2):
        lines.append("method w")
    lines += ["save out.vmg", "begin layers", f"  sor {float(rng.choice([1.0, 1.3, 1.5, 1.6]))}",
              f"  tol {float(rng.choice([1e-3, 3e-4, 1e-4]))}", f"  v-steps {int(rng.integers(1, 4))}",
              f"  mesh-size {n[0]} {n[1]} {n[2]}", "  bc all dirichlet"]
    for ax in "abc":
        k = int(rng.integers(0, 4))
        if k:
            lines.append(f"  bc {ax} {['', 'periodic', 'neumann', 'dirichlet'][k]}")
    lines += [f"  interp {'half' if rng.integers(0, 2) else 'full'}", "end layers"]
    for b in range(nbox):
        llc = [float(rng.uniform(0.05, 0.6)) * c for c in cell]
        ext = [float(rng.uniform(0.08, 0.35)) * c for c in cell]
        lines += ["begin box", f"  llc {llc[0]:.4f} {llc[1]:.4f} {llc[2]:.4f}", f"  cell {ext[0]:.4f} {ext[1]:.4f} {ext[2]:.4f}"]
        if b == 0 or rng.integers(0, 2):
            lines.append(f"  value {float(rng.choice([1.0, -1.0, 0.5]))}")
        else:
            lines += [f"  rho {float(rng.choice([3.0, 0.5, 8.0]))}", "  constant false"]
        lines.append("end box")
    return "\n".join(lines) + "\n"


def one_solve(ctx_unused, rng, idx):
    """a whole solve (gs_bottom_up or gs_w with its V cycles, m_gs_CDS.F90) of a random input file: this package's host
    over the library against the oracle's solve in the same order with the same exact sum; sweep counts, history of
    every convergence test and the field, bit for bit"""
    from multigrid_amd import cycle, mgio
    from tests.helpers import oracle_levels
    text = random_input(rng)
    dtype = np.float64 if rng.integers(0, 2) else np.float32
    kernel = int(rng.choice([0, 1, 2]))
    opts = dict(device_loop=bool(rng.integers(0, 2)), print_sums=bool(rng.integers(0, 2)))
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=dtype)
    method = mgio.read_method(IO)
    desc = dict(idx=idx, dtype=np.dtype(dtype).name, kernel=kernel, opts=opts, text=text)
    if method == cycle.CDS_W_CYCLE and len(list(top.levels())) < 2:
        STATS["not runnable"] += 1
        return                             # gs_w needs a child (m_gs_CDS.F90:118): not an input the reference can run
    try:
        V, hist, per = orc.solve(oracle_levels(top), method, dtype, order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT, max_sweeps=4000)
        stopped = bool(np.isnan(hist["sum"]).any())
    except orc.SolveStopped:
        stopped = True
    s = cycle.Solver(top, max_sweeps=4000, **opts)
    try:
        s.ctx.set_kernel(kernel)
        if stopped:
            try:
                s.mg_gs_cds(method)
            except Exception:
                STATS["stopped"] += 1
                return                     # a diverging or endless iteration is reported, not compared
            raise AssertionError(("the oracle stopped, the library did not", desc))
        s.mg_gs_cds(method)
        h = s.history_array()
        keep = hist if opts["print_sums"] else hist[hist["kind"] != 1]
        assert [l.itt for l in top.levels()] == per, ("counts", [l.itt for l in top.levels()], per, desc)
        assert len(h) == len(keep), ("history length", len(h), len(keep), desc)
        for f in ("layer", "kind", "itt", "tol", "sum", "err"):
            assert np.array_equal(h[f], keep[f]), (f, desc)
        assert bits_equal(top.dev.download(), V), ("field", desc)
        STATS["compared"] += 1
    finally:
        s.close()


def one_box_place(ctx, rng, idx):
    """grid_add_box's search for the cells a box holds (t_mg.f90:359-395) on the device against the oracle: skewed
    and orthogonal lattices, boxes inside, across and outside the cell, corners exactly on lattice points"""
    n = tuple(int(rng.integers(1, 70)) for _ in range(3))
    if rng.integers(0, 2):
        cell = np.diag(rng.uniform(1.0, 4.0, 3)) + rng.uniform(-0.4, 0.4, (3, 3))
    else:
        cell = np.diag([float(d) * float(rng.choice([0.1, 0.125, 0.3])) for d in n])
    dL = cell / np.array(n, dtype=np.float64)[None, :]
    off = rng.uniform(-0.5, 0.5, 3) if rng.integers(0, 2) else np.zeros(3)
    ext = np.abs(cell).sum(axis=1)
    for _ in range(8):
        if rng.integers(0, 2):       # corners on lattice points: the comparisons of the search meet equality
            i0 = np.array([int(rng.integers(-2, d + 2)) for d in n], dtype=np.float64)
            i1 = np.array([int(rng.integers(0, d + 3)) for d in n], dtype=np.float64)
            llc = off + dL @ i0
            bcell = dL * i1[None, :]
        else:
            llc = off + rng.uniform(-0.6, 0.9, 3) * ext
            bcell = np.diag(rng.uniform(-0.3, 0.7, 3) * ext) + rng.uniform(-0.05, 0.05, (3, 3))
        got, want = ctx.box_place(n, dL, off, llc, bcell), orc.box_place(n, dL, off, llc, bcell)
        assert np.array_equal(got, want), ("box_place", idx, n, dL.tolist(), off.tolist(), llc.tolist(), bcell.tolist(), got, want)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    global one_case
    if len(sys.argv) > 3 and sys.argv[3] == "transfers":
        one_case = one_transfer
    if len(sys.argv) > 3 and sys.argv[3] == "boxes":
        one_case = one_box_place
    if len(sys.argv) > 3 and sys.argv[3] == "solves":
        one_case = one_solve
    if len(sys.argv) > 3 and sys.argv[3] == "slabs":
        import torch  # noqa: F401  before the HIP library: one HIP runtime per process
        one_case = one_slab_case
    rng = np.random.default_rng(seed)
    ctx = lib.Context(0)
    t0 = time.time()
    try:
        for i in range(cases):
            one_case(ctx, rng, i)
            if i % 10 == 9 or one_case is one_solve:
                print(f"{i + 1} cases ok, {time.time() - t0:.0f} s", flush=True)
    finally:
        ctx.close()
    print(f"all {cases} cases identical to the oracle (seed {seed})" + (f" {STATS}" if one_case is one_solve else ""))


if __name__ == "__main__":
    main()
