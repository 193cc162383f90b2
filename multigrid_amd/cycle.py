"""Cycle control on the host: the reference's `m_gs_CDS` (src/m_gs_CDS.F90:21-315) driving the HIP
level operators through the C ABI.

Control flow, convergence tests and the arithmetic of the tests (real(grid_p)) follow the
reference line by line; what differs is below the boundary: every level lives in HBM for the
whole solve (grid_bring_back zero-fills instead of allocating, grid_hold_back keeps the storage),
the sweep is red-black, and grid_sum is the order-independent exact sum.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from . import grid as G
from . import lib

CDS_BOTTOM_UP, CDS_W_CYCLE = 1, 2          # m_gs_CDS.F90:12-13

HIST_DTYPE = np.dtype([("layer", "i4"), ("kind", "i4"), ("itt", "i4"), ("pad_", "i4"),
                       ("tol", "f8"), ("sum", "f8"), ("err", "f8")])


RB_SOR_LIMIT = 1.70   # largest over-relaxation at which the red-black v-cycle converged on every input tried


def divergence_hint(sor: float) -> str:
    """what to tell the user when a cycle blows up: the known stability limit of the red-black ordering"""
    if sor <= RB_SOR_LIMIT:
        return ""
    return (f"  The level runs sor = {sor:g}.  In red-black order the reference's cycle (restriction and prolongation "
            f"overwrite the levels, m_gs_CDS.F90:220-294) is stable only up to sor = 1.70 (periodic z) to 1.76 (Dirichlet "
            f"z), whereas the reference's lexicographic sweep tolerates its parser default of 1.8 (m_mg_io.F90:87): "
            f"put `sor 1.7` into the layers block, or sweep in the reference's order (`--reference-order`, "
            f"mgx_ctx_set_kernel(ctx, MGX_KERNEL_LEX): slower, the reference's results bit for bit).")


ORDER_AUTO, ORDER_RED_BLACK, ORDER_REFERENCE = "auto", "red-black", "reference"


class Solver:
    """Owns the device context and the resident levels of one hierarchy."""

    def __init__(self, top: G.MgGrid, device: int = 0, stream: Optional[int] = None, ctx: Optional[lib.Context] = None,
                 verbose: bool = False, print_sums: bool = True, max_sweeps: int = 1 << 24, device_loop: bool = True,
                 use_graphs: bool = False, order: str = ORDER_AUTO):
        if order not in (ORDER_AUTO, ORDER_RED_BLACK, ORDER_REFERENCE):
            raise ValueError(f"order = {order!r}")
        # sweep order: "red-black" (the fast kernels), "reference" (gs's own order, m_gs_CDS.F90:317-345: the
        # reference's results bit for bit at any over-relaxation, slower), "auto" = red-black unless a V cycle is asked
        # for with an over-relaxation beyond what that ordering takes (the parser's default 1.8 among them,
        # m_mg_io.F90:87): every input the reference solves must solve here
        self.order = order
        self.top = top
        self.ctx = ctx if ctx is not None else lib.Context(device, stream)
        self.own_ctx = ctx is None
        self.verbose = verbose
        self.print_sums = print_sums       # evaluate the print-only sums of gs_V (m_gs_CDS.F90:238,245,269,275)
        self.max_sweeps = max_sweeps
        self.device_loop = device_loop     # grid_solve's loop through mgx_level_solve (test evaluated on the device)
        # one cycle of gs_V recorded into a HIP graph and replayed (only the plain solver: a decomposed one
        # communicates between the library calls).  Off by default: measured on MI355X / ROCm 7.2, replaying the ~60
        # small launches of a cycle as a graph is no faster than enqueueing them one by one (128^3: 352 ms vs 341 ms)
        self.use_graphs = use_graphs and type(self) is Solver
        self._graphs = {}
        self.history: List[tuple] = []
        self.total_sweeps = 0
        self.t = top.dtype.type
        for g in top.levels():
            self.attach(g)

    # -- device residency -------------------------------------------------------------------------
    def make_level(self, g: G.MgGrid):
        """the resident level behind g (a subclass may hand out a slab of it)"""
        return self.ctx.level(g.layer, [int(v) for v in g.n], g.dtype)

    def attach(self, g: G.MgGrid):
        if g.dev is None:
            g.dev = self.make_level(g)
        self.push_params(g)

    def push_params(self, g: G.MgGrid):
        g.dev.set_params(g.sor, [float(v) for v in g.a], g.bc6())
        g.dev.set_boxes(*G.box_table(g))

    def close(self):
        for g in self.top.levels():
            if g.dev is not None:
                g.dev.close()
                g.dev = None
        if self.own_ctx:
            self.ctx.close()

    # -- the reference's level routines, now thin -----------------------------------------------
    def say(self, *a):
        if self.verbose:
            print(*a)

    def bring_back(self, g):               # grid_bring_back, t_mg.f90:601-612
        g.dev.bring_back()

    def hold_back(self, g):                # grid_hold_back, t_mg.f90:590-599
        g.dev.release()

    def setup(self, g, init: bool):        # grid_setup, t_mg.f90:465-588
        g.dev.setup(init)

    def restriction(self, g):              # grid_restriction, t_mg_interp.f90:9-21 (into g.child)
        lib.restrict(g.dev, g.child.dev, g.RES_method)

    def prolongation(self, g):             # grid_prolongation, t_mg_interp.f90:23-35 (into g.parent)
        lib.prolong(g.dev, g.parent.dev, g.PRO_method)

    def grid_sum(self, g) -> float:        # grid_sum, t_mg.f90:794-814 → real(grid_p)
        return self.t(g.dev.abs_sum())

    def count(self, g) -> int:             # grid_non_constant_elem, t_mg.f90:816-834
        return g.dev.count_nonconst()

    def tolerance(self, g) -> float:       # grid_tolerance, t_mg.f90:857-868
        return self.t(G.grid_tolerance(g))

    def gs_steps(self, g, nsteps: int, want_sum: bool, fetch: bool = True, defer: bool = False, want_err: bool = True):
        """`nsteps` × gs_step (m_gs_CDS.F90:296-315); returns the grid_sum after the last one if asked.
        grid%err = sqrt(max dV^2) of the last sweep.  With fetch=False the sweeps are only enqueued: the level keeps
        its result record until it is swept again, and grid%err is read from it when somebody asks (level_err).
        want_err=False (only with fetch=False): nobody will read grid%err of these sweeps before the level is swept
        again (an inner leg of a cycle): the sweeps run without reductions and grid%err keeps its last value."""
        if nsteps <= 0:
            return self.grid_sum(g) if want_sum else None
        g.itt += nsteps
        self.total_sweeps += nsteps
        if self.total_sweeps > self.max_sweeps:
            raise G.MgStop(f"more than {self.max_sweeps} sweeps")
        if not want_err and not want_sum and not fetch:
            g.dev.sweep(nsteps, 0)
            return None
        g.dev.sweep(nsteps, lib.RED_ERR | (lib.RED_SUM if want_sum else 0))
        if not (fetch or want_sum) or defer:     # defer: the caller reads the result later (fetch_deferred)
            g.err_pending = True
            return None
        return self.fetch_result(g, want_sum)

    def fetch_result(self, g, want_sum: bool):
        """read back the reductions of the level's last sweep: grid%err, and the grid_sum if asked"""
        e2, s, _ = g.dev.result()
        g.err = float(np.sqrt(self.t(e2)))
        g.err_pending = False
        if want_sum and s != s:
            # the reference would carry the overflowing sum through its `do while (tol > ...)` tests, which a NaN
            # ends as if converged (m_gs_CDS.F90:200-210); a diverging iteration is reported instead
            raise G.MgStop(f"layer {g.layer}: |V| reached 2^20 or is not finite after {g.itt} sweeps: the iteration diverges."
                           + divergence_hint(float(g.sor)))
        return self.t(s) if want_sum else None

    def level_err(self, g) -> float:
        """grid%err, read back from the level if its last sweeps were only enqueued"""
        if getattr(g, "err_pending", False):
            e2, _, _ = g.dev.result()
            g.err = float(np.sqrt(self.t(e2)))
            g.err_pending = False
        return g.err

    def record(self, g, kind, tol, s):
        self.history.append((g.layer, kind, g.itt, 0, float(tol), float(s), float(self.level_err(g))))

    def history_array(self) -> np.ndarray:
        return np.array(self.history, dtype=HIST_DTYPE)

    # -- m_gs_CDS -----------------------------------------------------------------------------------
    reference_order_available = True       # (a decomposed solver has no kernel for the reference's order)

    def choose_order(self, method: int):
        """select the sweep kernel for a solve by `method` (see `order` in __init__)"""
        if self.order == ORDER_RED_BLACK:
            return
        want = self.order == ORDER_REFERENCE
        if not want and method == CDS_W_CYCLE:
            high = [float(g.sor) for g in self.top.levels() if g.enabled and float(g.sor) > RB_SOR_LIMIT]
            if high:
                want = self.reference_order_available
                import sys
                if want:
                    print(f"multigrid_amd: `method v-cycle` with sor = {max(high):g} > {RB_SOR_LIMIT:g}: sweeping in the "
                          "reference's own order (its results bit for bit; the red-black kernels are stable up to "
                          f"sor {RB_SOR_LIMIT:g} and several times faster)", file=sys.stderr)
                else:
                    print("warning: `method v-cycle` with sor > 1.7." + divergence_hint(max(high)), file=sys.stderr)
        if want:
            if not self.reference_order_available:
                raise G.MgStop("the reference's sweep order needs undivided levels")
            self.ctx.set_kernel(lib.KERNEL_LEX)

    def mg_gs_cds(self, method: int = CDS_BOTTOM_UP, init: Optional[bool] = None):
        """m_gs_CDS.F90:21-57"""
        top = self.top
        self.choose_order(method)
        if init is None or init:
            self.bring_back(top)
            self.setup(top, True)
        if method == CDS_W_CYCLE:
            self.gs_w()
        elif method == CDS_BOTTOM_UP:
            self.gs_bottom_up()
        else:
            raise G.MgStop("Error on method used for CDS")

    def gs_bottom_up(self):
        """m_gs_CDS.F90:59-104"""
        grid = self.top
        while grid.child is not None:
            grid = grid.child
            if grid.child is None:
                self.bring_back(grid)
                self.setup(grid, False)
        while grid is not None:
            old_itt = grid.itt
            self.grid_solve(grid)
            self.say(f"Completed ({grid.layer}) cycle in {grid.itt - old_itt} cycles")
            if grid.parent is not None:
                self.bring_back(grid.parent)
                self.prolongation(grid)
                self.hold_back(grid)
            grid = grid.parent

    def grid_solve(self, grid):
        """m_gs_CDS.F90:181-218"""
        if not grid.enabled:
            return
        t = self.t
        nr = t(1) / t(self.count(grid))
        gtol = self.tolerance(grid)
        tol = t(gtol + t(1))
        old_sum = self.grid_sum(grid)
        self.say(f"Initial sum: {float(old_sum * nr):10.7f}")
        old_itt = grid.itt
        if self.device_loop and self.solve_on_device(grid, nr, gtol, old_sum, old_itt):
            return
        while tol > gtol:
            new_sum = self.gs_steps(grid, 1, True)
            tol = t(abs(t(old_sum - new_sum)) * nr)
            old_sum = new_sum
            self.record(grid, 0, tol, new_sum)
            self.say(f" itt {grid.itt - old_itt:6d} {float(tol):10.7f} {float(new_sum * nr):10.7f} {self.level_err(grid):10.7f}")

    def solve_on_device(self, grid, nr, gtol, old_sum, old_itt) -> bool:
        """the loop of grid_solve (m_gs_CDS.F90:200-216) with the convergence test evaluated on the device: the same
        sweeps, history and prints, one host wait per batch of sweeps.  False = the level does not offer it (a slab, a
        test double) or the library does not take it (more than 32 boxes): the caller loops itself."""
        solve = getattr(grid.dev, "solve", None)
        if solve is None:
            return False
        t = self.t
        # one call takes at most SOLVE_CHUNK sweeps (its history array is sized for them); a level that needs more is
        # continued from the state the call left (old_sum = the last sum), like the loop of grid_solve itself, which
        # has no cap (m_gs_CDS.F90:200)
        while True:
            budget = self.max_sweeps - self.total_sweeps
            if budget <= 0:
                raise G.MgStop(f"more than {self.max_sweeps} sweeps")
            try:
                hist, n, status = solve(float(gtol), self.count(grid), float(old_sum), min(budget, self.SOLVE_CHUNK))
            except lib.MgxError as e:
                if e.code == -4:              # MGX_ESTATE: not on this level
                    return False
                raise
            assert n == len(hist)
            for tol_, sum_, e2 in hist:
                grid.itt += 1
                self.total_sweeps += 1
                grid.err = float(np.sqrt(t(e2)))
                grid.err_pending = False
                if sum_ != sum_:
                    break
                self.record(grid, 0, t(tol_), t(sum_))
                old_sum = t(sum_)
                self.say(f" itt {grid.itt - old_itt:6d} {float(t(tol_)):10.7f} {float(t(sum_) * nr):10.7f} {grid.err:10.7f}")
            if status == 2:
                raise G.MgStop(f"layer {grid.layer}: |V| reached 2^20 or is not finite after {grid.itt} sweeps: the iteration diverges."
                               + divergence_hint(float(grid.sor)))
            if status == 0:
                return True

    SOLVE_CHUNK = 1 << 16

    def _leg(self, grid, want_sum: bool = False, defer: bool = False):
        """steps × gs_step between two (print-only) sums: m_gs_CDS.F90:237-248 / :268-278.  Returns the grid_sum
        after the last step when asked (the caller's own grid_sum would read the same field)."""
        t = self.t
        if self.print_sums:
            nr = t(1) / t(self.count(grid))
            old_sum = self.grid_sum(grid)
            new_sum = self.gs_steps(grid, grid.steps, True)
            tol = t(abs(t(old_sum - new_sum)) * nr)
            self.record(grid, 1, tol, new_sum)
            self.say(f"      {'  ' * grid.layer}{grid.steps:6d} {float(tol):10.7f} {float(new_sum * nr):10.7f}")
            return new_sum
        # nobody reads the residual of an inner leg (the reference evaluates it only to print it): its sweeps are
        # enqueued without reductions and without waiting for them; the last leg of a cycle carries both
        return self.gs_steps(grid, grid.steps, want_sum, fetch=False, defer=defer, want_err=want_sum)

    def gs_V(self, pg, cg):
        """m_gs_CDS.F90:220-294.  A cycle holds no host decision (when the print-only sums are off), so from its
        second run on a pair of levels it is recorded once into a HIP graph and then replayed."""
        if not (self.use_graphs and not self.print_sums and not self.verbose and hasattr(self.ctx, "capture_begin")):
            return self._gs_V_calls(pg, cg)
        levels = list(self.top.levels())
        key = (pg.layer, cg.layer)
        st = self._graphs.get(key)
        if st is None:                       # first cycle of this pair: run it, so that every tile list and table exists
            self._graphs[key] = {"graphs": [], "failed": False}
            return self._gs_V_calls(pg, cg)
        if st["failed"]:
            return self._gs_V_calls(pg, cg)
        for gr, delta, dtot in st["graphs"]:  # a graph replays from the buffer state it was recorded in
            try:
                gr.launch()
            except lib.MgxError as e:
                if e.code != -4:
                    raise
                continue
            for g, d in zip(levels, delta):
                g.itt += d
            self.total_sweeps += dtot
            break
        else:
            if len(st["graphs"]) >= 4:       # states keep coming that no recording starts from: call by call
                st["failed"] = True
                return self._gs_V_calls(pg, cg)
            before, tot0 = [g.itt for g in levels], self.total_sweeps
            try:
                self.ctx.capture_begin()
                try:
                    self._gs_V_calls(pg, cg, defer=True)
                except BaseException:
                    self.ctx.capture_end(keep=False)
                    raise
                gr = self.ctx.capture_end()
            except lib.MgxError as e:
                if e.code != -4:             # MGX_ESTATE: this sequence cannot be recorded: call by call
                    raise
                for g, b in zip(levels, before):
                    g.itt = b
                self.total_sweeps = tot0
                st["failed"] = True
                return self._gs_V_calls(pg, cg)
            delta = [g.itt - b for g, b in zip(levels, before)]
            st["graphs"].append((gr, delta, self.total_sweeps - tot0))
            gr.launch()                      # the counters of this cycle are in: it is the recording's first run
        if self.total_sweeps > self.max_sweeps:
            raise G.MgStop(f"more than {self.max_sweeps} sweeps")
        for g, d in zip(levels, st["graphs"][0][1]):
            if d:
                g.err_pending = True
        return self.fetch_result(pg, True) if pg.steps > 0 else None

    def _gs_V_calls(self, pg, cg, defer: bool = False):
        """m_gs_CDS.F90:220-294.  Returns grid_sum(pg) as it stands when the cycle ends (the sum the caller
        evaluates next, m_gs_CDS.F90:149, comes out of the last sweep's own reduction), or None."""
        grid = pg
        while grid.layer != cg.layer:
            if not grid.child.enabled:
                break
            self._leg(grid)
            self.bring_back(grid.child)
            self.restriction(grid)
            self.hold_back(grid)
            grid = grid.child
        while True:
            last = grid.layer == pg.layer
            s_end = self._leg(grid, want_sum=last, defer=defer)
            if last:
                return s_end if grid.steps > 0 else None
            if grid.parent is not None:
                self.bring_back(grid.parent)
                self.prolongation(grid)
                self.hold_back(grid)
            grid = grid.parent

    def gs_w(self):
        """m_gs_CDS.F90:106-179.  `otol` aliases a variable assigned once (:129) until the top level has
        been reached, and `itol` from then on (:162-166)."""
        t = self.t
        top = self.top
        if top.child is None:
            raise G.MgStop("gs_w needs at least two levels (the reference dereferences pg%child%child)")
        pg = cg = top
        while pg.child.child is not None:
            if not pg.child.child.enabled:
                break
            self.bring_back(pg.child)
            self.restriction(pg)
            self.hold_back(pg)
            pg = pg.child
            cg = pg.child
        tol = t(self.tolerance(pg) + t(1))
        itol = t(0)
        alias_itol = False
        top_tol = self.tolerance(top)
        while (itol if alias_itol else tol) > top_tol:
            nr = t(t(1) / t(self.count(pg))) / t(pg.steps)
            old_sum = self.grid_sum(pg)
            old_itt = pg.itt
            self.say(f"Running between {pg.layer} and {cg.layer} in {pg.steps} steps.")
            pg_tol = self.tolerance(pg)
            itol = t(pg_tol + t(1))
            while itol > pg_tol:
                new_sum = self.gs_V(pg, cg)
                if new_sum is None:
                    new_sum = self.grid_sum(pg)
                itol = t(abs(t(old_sum - new_sum)) * nr)
                old_sum = new_sum
                self.record(pg, 2, itol, new_sum)
                self.say(f" itt {pg.itt - old_itt:6d} {float(itol):10.7f} {float(new_sum * nr):10.7f} {self.level_err(pg):10.7f}")
            self.say(f"Completed: {pg.layer}:{cg.layer} in {pg.itt - old_itt} itt. per. lvl")
            if pg.layer == top.layer:
                cg = pg
                if not alias_itol:
                    itol = tol
                    alias_itol = True
            if pg.parent is not None:
                self.bring_back(pg.parent)
                self.prolongation(pg)
                self.hold_back(pg)
                cg = pg
                pg = pg.parent


def mg_gs_cds(top: G.MgGrid, method: int = CDS_BOTTOM_UP, init: Optional[bool] = None, **kw) -> Solver:
    """convenience: build a Solver, run the reference's entry point, return the solver (history, levels)"""
    s = Solver(top, **kw)
    s.mg_gs_cds(method, init)
    return s
