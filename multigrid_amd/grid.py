"""Host-side level hierarchy: the reference's `t_mg` data model (src/t_mg.f90) in Python.

Same names, argument meaning and error behaviour as the Fortran routines, so that code written
against the reference's library API (test/test2..6.f90) reads the same here.  Only host
bookkeeping lives in this module; every field operation goes to the HIP library.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# src/t_bc.f90:6-15
MG_BC_A0, MG_BC_A1, MG_BC_B0, MG_BC_B1, MG_BC_C0, MG_BC_C1 = 1, 2, 4, 8, 16, 32
MG_BC_PERIODIC, MG_BC_DIRICHLET, MG_BC_NEUMANN = 1, 2, 3
# src/t_mg.f90:11-20
MG_CELL_A0, MG_CELL_A1, MG_CELL_B0, MG_CELL_B1, MG_CELL_C0, MG_CELL_C1 = 1, 2, 4, 8, 16, 32
MG_INTERP_FULL, MG_INTERP_HALF = 1, 2

HUGE_INT = 2147483647                      # huge(0)
PI = 3.1415926535897932384626433832795028  # src/m_unit.f90:8
ANG = 1.0 / 0.529177                       # src/m_unit.f90:5


class MgStop(RuntimeError):
    """the reference's `stop 'message'` (t_mg.f90:135,334,344,352; m_gs_CDS.F90:53)"""


@dataclass
class MgBox:
    """t_mg.f90:62-72"""
    place: np.ndarray = field(default_factory=lambda: np.zeros(6, dtype=np.int64))  # xmin,xmax,ymin,ymax,zmin,zmax
    val: float = 0.0
    rho: float = 1.0
    constant: bool = False

    def delete(self):                       # delete_box, t_mg.f90:690-696
        self.place[:] = 0
        self.val = 0.0
        self.rho = 1.0
        self.constant = False


class MgGrid:
    """type(mg_grid), t_mg.f90:22-60.  `dtype` plays the role of grid_p (t_mg.f90:8)."""

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.enabled = True
        self.offset = np.zeros(3)
        self.cell = np.zeros((3, 3))        # cell[:, i] = i-th lattice vector (Fortran cell(:,i))
        self.dL = np.zeros((3, 3))
        self.dLL = np.zeros(3)
        self.dVol = 0.0
        self.Vol = 0.0
        self.n = np.zeros(3, dtype=np.int64)
        self.sor = 0.0
        self.a = np.zeros(3, dtype=self.dtype)
        self.tol = 0.0
        self.itt = 0
        self.steps = 2
        self.layer = 0
        self.parent: Optional["MgGrid"] = None
        self.child: Optional["MgGrid"] = None
        self.N_box = 0
        self.box: List[MgBox] = []
        self.err = 0.0
        self.BC = np.full((2, 3), MG_BC_DIRICHLET, dtype=np.int64)   # BC[k, j]: side k of axis j (t_bc.f90:20)
        self.PRO_method = MG_INTERP_FULL
        self.RES_method = MG_INTERP_FULL
        self.dev = None                      # resident device level (multigrid_amd.lib.DeviceLevel)

    # -- conveniences with no Fortran counterpart -------------------------------------------------
    def bc6(self):
        """A0,A1,B0,B1,C0,C1"""
        return [int(self.BC[0, 0]), int(self.BC[1, 0]), int(self.BC[0, 1]), int(self.BC[1, 1]),
                int(self.BC[0, 2]), int(self.BC[1, 2])]

    def levels(self):
        g, out = self, []
        while g is not None:
            out.append(g)
            g = g.child
        return out

    def _r(self, v: float) -> float:
        """a value after assignment to a real(grid_p) variable"""
        return float(self.dtype.type(v))


def init_grid(grid: MgGrid, n: Sequence[int], cell, boxes: int, tol: Optional[float] = None, offset=None,
              sor: Optional[float] = None, steps: Optional[int] = None) -> None:
    """t_mg.f90:76-151"""
    delete_grid(grid)
    grid.steps = 2 if steps is None else int(steps)
    grid.n = np.array([int(v) for v in n], dtype=np.int64)
    cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
    for i in range(3):
        grid.dL[:, i] = cell[:, i] / grid.n[i]
        grid.dLL[i] = grid.dL[i, :].sum()

    def det(c):
        return ((c[1, 0] * c[2, 1] - c[2, 0] * c[1, 1]) * c[0, 2] + (c[2, 0] * c[0, 1] - c[0, 0] * c[2, 1]) * c[1, 2]
                + (c[0, 0] * c[1, 1] - c[1, 0] * c[0, 1]) * c[2, 2])
    grid.Vol = det(cell)
    grid.dVol = det(grid.dL)
    grid.enabled = True
    grid.layer = grid.parent.layer + 1 if grid.parent is not None else 1
    grid.offset = np.zeros(3) if offset is None else np.asarray(offset, dtype=np.float64).copy()
    for i in range(3):
        grid.cell[:, i] = grid.dL[:, i] * grid.n[i]
    grid.itt = 0
    if tol is not None:
        grid.tol = float(tol)
    elif grid.parent is not None:
        grid.tol = grid.parent.tol
    else:
        raise MgStop("Tolerance not set")
    grid.sor = 2.0 / (1.0 + PI / int(grid.n.max()))
    if sor is not None:
        grid.sor = float(sor)
    grid.a = np.full(3, 1.0 / 3.0).astype(grid.dtype)
    grid.N_box = int(boxes)
    grid.box = [MgBox() for _ in range(grid.N_box)]


def g2g(n1: int, i1: int, n2: int) -> int:
    """index i1 of a grid with n1 cells seen from a grid with n2 cells along the same direction
    (t_mg_interp.f90:995-999: integer division, clamped to 1..n2); the transfer kernels' index tables hold this map"""
    return max(1, min(int(i1) * int(n2) // int(n1), int(n2)))


def _new_grid_size(grid: MgGrid):
    """t_mg.f90:183-201"""
    n = [int(v) // 2 for v in grid.n]
    for i in range(3):
        if n[i] < 20:
            if any(v > n[i] * 3 for v in n):
                n[i] = int(grid.n[i])
            else:
                return [0, 0, 0]
    return n


def init_grid_children_half(top: MgGrid, max_layer: Optional[int] = None) -> None:
    """t_mg.f90:153-203"""
    lmax = top.layer - 1 if max_layer is None else int(max_layer)
    n = _new_grid_size(top)
    grid = top
    while all(v != 0 for v in n) and lmax != grid.layer:
        tmp = MgGrid(top.dtype)
        grid.child = tmp
        tmp.parent = grid
        init_grid(tmp, n, grid.cell, grid.N_box, tol=grid.tol, offset=grid.offset, sor=grid.sor)
        grid = tmp
        n = _new_grid_size(grid)


def grid_set(grid: MgGrid, sor=None, tol=None, layer=None, weight=None, offset=None, steps=None, restrict=None,
             prolong=None) -> None:
    """t_mg.f90:205-278"""
    g = grid
    if layer is not None:
        while layer != g.layer:
            g = g.child
            if g is None:
                return
    if sor is not None:
        g.sor = float(sor)
    if tol is not None:
        g.tol = float(tol)
    if steps is not None:
        g.steps = int(steps)
    if offset is not None:
        g.offset = np.asarray(offset, dtype=np.float64).copy()
    if restrict is not None:
        g.RES_method = int(restrict)
    if prolong is not None:
        g.PRO_method = int(prolong)
    if weight is not None:
        celll = np.array([(g.dL[:, i] ** 2).sum() for i in range(3)])
        if weight == 0:
            a = np.ones(3)
        elif weight == 1:
            a = np.array([celll[1] * celll[2], celll[0] * celll[2], celll[0] * celll[1]])
        elif weight == -1:
            a = np.array([celll[0] / (celll[1] * celll[2]), celll[1] / (celll[0] * celll[2]),
                          celll[2] / (celll[0] * celll[1])])
        else:
            a = g.a.astype(np.float64)
        a = a.astype(g.dtype)
        s = g.dtype.type(a[0] + a[1])
        s = g.dtype.type(s + a[2])
        g.a = (a / s).astype(g.dtype)


def grid_BC(grid: MgGrid, BC: int, plane: Optional[int] = None) -> None:
    """t_mg.f90:280-321 (recursive over the children)"""
    if BC not in (MG_BC_PERIODIC, MG_BC_DIRICHLET, MG_BC_NEUMANN):
        return
    lplane = 63 if plane is None else int(plane)
    for bit, (k, j) in ((MG_CELL_A0, (0, 0)), (MG_CELL_A1, (1, 0)), (MG_CELL_B0, (0, 1)), (MG_CELL_B1, (1, 1)),
                        (MG_CELL_C0, (0, 2)), (MG_CELL_C1, (1, 2))):
        if lplane & bit == bit:
            grid.BC[k, j] = BC
    if grid.child is not None:
        grid_BC(grid.child, BC, plane=plane)


# how grid_add_box finds the cells of a box: "auto" = per-axis arithmetic on the host for orthogonal cells, the
# device kernel for skewed ones; "device" = the device kernel always
BOX_PLACEMENT = os.environ.get("MGX_BOX_PLACEMENT", "auto")
PLACEMENT_CONTEXT = None     # lib.Context that places boxes on skewed cells (None: lib.shared_context() of the rank's device)


class placement_context:
    """`with grid.placement_context(ctx): top = mgio.iomg_read(...)`: boxes on skewed cells are placed through `ctx`
    (its device, its stream) instead of a context the package opens by itself"""

    def __init__(self, ctx):
        self.ctx, self.old = ctx, None

    def __enter__(self):
        global PLACEMENT_CONTEXT
        self.old, PLACEMENT_CONTEXT = PLACEMENT_CONTEXT, self.ctx
        return self.ctx

    def __exit__(self, *exc):
        global PLACEMENT_CONTEXT
        PLACEMENT_CONTEXT = self.old
        return False


def grid_add_box(grid: MgGrid, llc, box_cell, val: float, rho: float, constant: bool, recurse: Optional[bool] = None) -> None:
    """t_mg.f90:323-435: cell centres inside [llc, urc] (component-wise) define the index range."""
    if grid.N_box == 0:
        raise MgStop("No boxes")
    box = None
    for b in grid.box:
        if not np.all(b.place == 0):
            continue
        box = b
        break
    if box is None:
        raise MgStop("No boxes available")
    box.delete()
    box.val = grid._r(val)
    box.rho = grid._r(rho)
    if rho < 0.0:
        raise MgStop("not available")
    box.constant = bool(constant)
    llc = np.asarray(llc, dtype=np.float64)
    box_cell = np.asarray(box_cell, dtype=np.float64).reshape(3, 3)
    lo = np.full(3, HUGE_INT, dtype=np.int64)
    hi = np.zeros(3, dtype=np.int64)
    offset = grid.offset + (grid.dL[:, 0] + grid.dL[:, 1] + grid.dL[:, 2]) * 0.5
    urc = llc + box_cell.sum(axis=1)

    n1, n2, n3 = (int(v) for v in grid.n)
    diagonal = all(grid.dL[i, j] == 0.0 for i in range(3) for j in range(3) if i != j)
    if diagonal and BOX_PLACEMENT != "device":
        # component i of a cell centre depends on the index along axis i only (adding the zero
        # off-diagonal terms is exact), so both inclusion tests separate per axis: O(n1 + n2 + n3) on the host
        cs = [grid.dL[i, i] * np.arange(n, dtype=np.float64) + offset[i] for i, n in enumerate((n1, n2, n3))]
        fwd = [llc[i] <= cs[i] for i in range(3)]
        inside = [np.nonzero(fwd[i] & (cs[i] <= urc[i]))[0] for i in range(3)]
        if all(len(ix) for ix in inside):                       # all(llc <= xyz) .and. all(xyz <= urc)
            for i in range(3):
                lo[i], hi[i] = inside[i][0] + 1, inside[i][-1] + 1
        mirrored = [np.nonzero((urc[i] <= cs[i]) & (cs[i] <= llc[i]))[0] for i in range(3)]
        if all(len(ix) for ix in mirrored):                     # the `else if` branch, t_mg.f90:378-385
            for x in mirrored[0]:
                for y in mirrored[1]:
                    for z in mirrored[2]:
                        if fwd[0][x] and fwd[1][y] and fwd[2][z]:
                            continue                            # taken by the first branch
                        for i, v in enumerate((x, y, z)):
                            lo[i] = min(lo[i], v + 1)
                            hi[i] = max(hi[i], v + 1)
    else:
        # skewed cells: the reference's loop over every cell (t_mg.f90:363-390), run on the GPU (mgx_box_place).
        # There is no host form of it on purpose: this package has no CPU fallback for anything the library does.
        # The context: the one the caller named (placement_context), else one on the rank's own device.
        from . import lib
        ctx = PLACEMENT_CONTEXT if PLACEMENT_CONTEXT is not None else lib.shared_context()
        pl = ctx.box_place((n1, n2, n3), grid.dL, grid.offset, llc, box_cell)
        lo[:] = pl[0::2]
        hi[:] = np.where(pl[0::2] == HUGE_INT, 0, pl[1::2])

    hi = np.maximum(hi, lo)                  # "ensure at least one point", t_mg.f90:392-395
    box.place[:] = [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]
    if np.all(lo == HUGE_INT) and np.all(hi == 0):     # never true after the max above: the slot stays taken
        box.delete()
    if grid.child is not None and (recurse is None or recurse):
        grid_add_box(grid.child, llc, box_cell, val, rho, constant, recurse=recurse)


def grid_add_point(grid: MgGrid, llc, val, rho, constant) -> None:
    """t_mg.f90:437-448"""
    grid_add_box(grid, llc, np.zeros((3, 3)), val, rho, constant)


def grid_add_line(grid: MgGrid, llc, dir: int, l: float, val, rho, constant) -> None:
    """t_mg.f90:450-463 (dir is 1-based)"""
    bl = np.zeros((3, 3))
    bl[dir - 1, dir - 1] = l
    grid_add_box(grid, llc, bl, val, rho, constant)


def delete_grid(grid: MgGrid) -> None:
    """t_mg.f90:614-632"""
    if grid.child is not None:
        delete_grid(grid.child)
        grid.child = None
    grid.box = []
    grid.N_box = 0
    if grid.dev is not None:
        grid.dev.close()
        grid.dev = None
    grid.layer = 0


def layers(grid: MgGrid, enabled: Optional[bool] = None) -> int:
    """t_mg.f90:712-729"""
    n, t = 1, grid.child
    while t is not None:
        if enabled is None or enabled == t.enabled:
            n += 1
        t = t.child
    return n


def grid_layer(grid: MgGrid, layer: int) -> MgGrid:
    """t_mg.f90:836-855"""
    lg, ll = grid, layer
    if ll < 1:
        ll = layers(grid, enabled=True) + 1 + ll
    if ll == 0:
        return lg
    while ll > 1:
        if lg.child is None:
            return lg
        lg = lg.child
        ll -= 1
    return lg


def grid_onoff_layer(in_grid: MgGrid, onoff: bool, layer: Optional[int] = None) -> None:
    """t_mg.f90:764-792"""
    llayer = -1 if layer is None else layer
    if llayer < 0:
        llayer = layers(in_grid) + 1 + llayer
    if llayer == 0:
        return
    g = in_grid
    while llayer > 1:
        g = g.child
        if g is None:
            return
        llayer -= 1
    g.enabled = bool(onoff)


def grid_delete_layer(in_grid: MgGrid, layer: Optional[int] = None) -> None:
    """t_mg.f90:731-762"""
    llayer = -1 if layer is None else layer
    if llayer < 0:
        llayer = layers(in_grid) + 1 + llayer
    if llayer == 0:
        return
    g = in_grid
    while llayer > 1:
        g = g.child
        if g is None:
            return
        llayer -= 1
    delete_grid(g)
    if g.parent is not None:
        g.parent.child = None
        g.parent = None


def grid_tolerance(grid: MgGrid) -> float:
    """t_mg.f90:857-868, in grid_p arithmetic"""
    t = grid.dtype.type
    fi = np.finfo(grid.dtype)
    vmax, vmin = t(-fi.max), t(fi.max)
    for b in grid.box:
        v = t(b.val)
        vmin = min(vmin, v)
        vmax = max(vmax, v)
    with np.errstate(over="ignore"):
        d = abs(t(vmax - vmin))
    return float(t(grid.tol * float(d)))


def box_table(grid: MgGrid):
    """the level's box slots as arrays for mgx_level_set_boxes"""
    place = np.array([b.place for b in grid.box], dtype=np.int64).reshape(-1, 6)
    place = np.clip(place, -HUGE_INT, HUGE_INT).astype(np.int32)
    val = np.array([b.val for b in grid.box], dtype=np.float64)
    rho = np.array([b.rho for b in grid.box], dtype=np.float64)
    const = np.array([int(b.constant) for b in grid.box], dtype=np.int32)
    return place, val, rho, const


def print_grid(top: MgGrid, file=None) -> None:
    """t_mg.f90:870-931 (layout only approximately: informational output)"""
    import sys
    out = file or sys.stdout
    print(" *******************************************", file=out)
    g = top
    while g is not None:
        ind = " " * (g.layer - 1)
        if g is top:
            print(f"{ind} -- Cell (Ang):", file=out)
            for j in range(3):
                print(ind + "    " + " ".join(f"{v / ANG:10.5f}" for v in g.cell[:, j]), file=out)
        print(f"{ind} -- Layer: {g.layer} size: {g.n[0]} {g.n[1]} {g.n[2]}", file=out)
        print(f"{ind} -- tolerance: {grid_tolerance(g):10.3E}", file=out)
        print(f"{ind} -- SOR: {g.sor:6.4f}", file=out)
        print(f"{ind} -- a(3): " + " ".join(f"{float(v):10.4E}" for v in g.a), file=out)
        names = {1: "periodic", 2: "dirichlet", 3: "neumann"}
        for j in range(3):
            print(f"{ind} -- BC({j + 1}) {{ {names[int(g.BC[0, j])]} : {names[int(g.BC[1, j])]} }}", file=out)
        for b in g.box:
            print(f"{ind} ++ Box {{ value: {b.val:10.3E} constant: {'T' if b.constant else 'F'} place: "
                  + " ".join(str(int(v)) for v in b.place) + " }", file=out)
        g = g.child
    print(" *******************************************", file=out)
