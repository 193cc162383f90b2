"""z-slab decomposition of one level over the ranks of a torch.distributed group
(one process per GPU; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference sweeps a level with OpenMP threads over one address space
(src/m_gs_CDS.F90:317-345); across GPUs the same red-black sweep needs exactly one exchange
per sweep: rank r owns the global planes z0+1 .. z0+n3 of the level and keeps two halo planes
on either side.  One fused sweep (black on planes 0 .. n3+1 from the old values, then red on
1 .. n3) is local; afterwards the new planes 1, 2 go to the lower neighbour's n3+1, n3+2 and
the new planes n3-1, n3 to the upper neighbour's -1, 0.  Whole padded planes travel, so the
frozen x/y ghost columns of the neighbour's planes come along.  The global z ghost planes
(0 and n3g+1) stay as grid_setup left them, like every ghost in the reference
(t_mg.f90:465-588 runs only at set-up); only a periodic z boundary needs the far rank's plane
at set-up time.

Reductions: max (vv-V)^2 is a MAX all-reduce; sum |V| is the exact integer sum
(units of 2^-72), sent as four 32-bit digits in int64 and added with a SUM all-reduce, so the
result has the same bits for any number of ranks.

The level object behind a slab only needs: setup(init), sweep(1, reduce), sweep_part(part, reduce),
result(), halo_written(), plane_tensor(z, count); multigrid_amd.lib.DeviceLevel provides them on the GPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

import struct

from . import grid as G
from . import lib

SUM_UNITS = 1 << 72


def split_planes(n3: int, world: int) -> List[Tuple[int, int]]:
    """(z0, n3_local) of every rank: contiguous, sizes differ by at most one plane"""
    base, rem = divmod(int(n3), int(world))
    out, z0 = [], 0
    for r in range(world):
        k = base + (1 if r < rem else 0)
        out.append((z0, k))
        z0 += k
    return out


class _DevicePlanes:
    """a run of planes of a device level as a zero-copy torch uint8 tensor (__cuda_array_interface__)"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_plane_tensor(lv: "lib.DeviceLevel", z: int, count: int):
    if not lib.torch_loaded_first:
        raise RuntimeError("import torch before the first multigrid_amd.lib.Context is created: the HIP library must "
                           "bind to the HIP runtime PyTorch ships, or torch cannot see the planes")
    import torch
    ptr, nbytes = lv.plane_ptr(z, count)
    cache = lv.__dict__.setdefault("_plane_tensors", {})      # a level has two buffers: a handful of distinct runs
    t = cache.get((ptr, nbytes))
    if t is None:
        t = cache[(ptr, nbytes)] = torch.as_tensor(_DevicePlanes(ptr, nbytes), device=torch.device("cuda", lv.ctx.device))
    return t


def check_stream_is_current(ctx):
    """RCCL enqueues its transfers on torch's current stream.  They are ordered against the library's kernels (planes
    written before they are sent, received before the next sweep reads them) only if the library enqueues on that
    very stream; a context with a stream of its own would race them."""
    import torch
    cur = torch.cuda.current_stream(torch.device("cuda", ctx.device)).cuda_stream
    mine = getattr(ctx, "stream", None)
    if not mine or int(mine) != int(cur):
        raise RuntimeError(
            "z slabs over RCCL need the mgx context on torch's current stream: create a torch.cuda.Stream, make it "
            "current (torch.cuda.set_stream) and pass its cuda_stream to multigrid_amd.lib.Context "
            f"(context stream {mine!r}, current stream {cur!r}; the default stream, handle 0, cannot be passed)")


class _StagedTransfers:
    """transfers of host copies of device planes (gloo): wait() also moves what was received onto the device"""

    def __init__(self, works, received):
        self.works, self.received = works, received

    def wait(self):
        import torch
        for w in self.works:
            w.wait()
        for planes, host in self.received:
            planes.copy_(host)
        torch.cuda.synchronize()


class SlabLevel:
    """one rank's slab of a level + the halo exchange and the all-reduced reductions"""

    def __init__(self, ctx, top: G.MgGrid, rank: int, world: int, group=None,
                 level_factory: Optional[Callable] = None, plane_tensor: Optional[Callable] = None, overlap: bool = True):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world, self.group = int(rank), int(world), group
        self.grid = top
        n = [int(v) for v in top.n]
        parts = split_planes(n[2], world)
        self.z0, self.n3 = parts[rank]
        if min(k for _, k in parts) < 2 and world > 1:
            raise ValueError(f"level of depth {n[2]} over {world} ranks: every slab needs at least 2 planes")
        self.n_global = tuple(n)
        make = level_factory if level_factory is not None else ctx.level
        self.lv = make(top.layer, n, top.dtype, self.z0, self.n3)
        self.lv.set_params(top.sor, [float(v) for v in top.a], top.bc6())
        self.lv.set_boxes(*G.box_table(top))
        self._plane_tensor = plane_tensor if plane_tensor is not None else device_plane_tensor
        self.lower = rank - 1 if rank > 0 else None
        self.upper = rank + 1 if rank < world - 1 else None
        self.periodic_z = top.bc6()[4] == lib.BC_PERIODIC or top.bc6()[5] == lib.BC_PERIODIC
        self.reduce = lib.RED_ERR | lib.RED_SUM
        self.overlap = overlap and hasattr(self.lv, "sweep_part")
        self.depth = 2

    def choose_depth(self, allow_pairs: bool = True):
        """halo planes per side: 2 (one sweep per exchange), or 4 where the library takes two sweeps in one pass over
        the slab (gs_V's legs: half as many exchanges and passes).  Every rank must exchange the same number of
        planes: the ranks agree (a slab thinner than four planes anywhere, or a rank whose library says no: 2)."""
        ask = getattr(self.lv, "pair_pass", None)
        yes = 1 if (allow_pairs and ask is not None and self.n3 >= 4 and ask()) else 0
        if self.world > 1:
            import torch
            t = torch.tensor([yes], dtype=torch.int64, device=self._scalar_device())
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
            yes = int(t.item())
        self.depth = 4 if yes else 2
        return self.depth

    # ---- communication -------------------------------------------------------------------------
    def planes(self, z: int, count: int):
        return self._plane_tensor(self.lv, z, count)

    def _stream_ordered(self) -> bool:
        """RCCL orders its transfers after the kernels enqueued on the current stream.  gloo (tests) reads the planes
        from the host as soon as a send is posted, so a device level has to be idle before one is."""
        ans = getattr(self, "_ordered", None)
        if ans is None:
            ctx = getattr(self.lv, "ctx", None)
            ans = ctx is None or self.world == 1 or self.dist.get_backend(self.group) != "gloo"
            if ans and ctx is not None and self.world > 1:
                check_stream_is_current(ctx)
            self._ordered = ans
        return ans

    def _scalar_device(self):
        """where the few numbers of a reduction live: the GPU for RCCL, the host for gloo"""
        if self.dist.get_backend(self.group) == "gloo":
            return "cpu"
        return self.planes(1, 1).device

    def _post(self, spec):
        """spec: (is_send, planes, peer) in the order both sides of a link agree on; returns objects with .wait().
        RCCL takes the device planes as they are, ordered on the stream.  gloo moves host memory: device planes are
        staged through host buffers (the level idle before they are read, received planes copied in by wait())."""
        if not spec:
            return []
        P, d = self.dist.P2POp, self.dist
        if self._stream_ordered():
            return d.batch_isend_irecv([P(d.isend if snd else d.irecv, t, peer, self.group) for snd, t, peer in spec])
        import torch
        self.lv.ctx.sync()
        torch.cuda.synchronize()
        ops, after = [], []
        for snd, t, peer in spec:
            h = t.cpu() if snd else torch.empty(t.shape, dtype=t.dtype)
            ops.append(P(d.isend if snd else d.irecv, h, peer, self.group))
            if not snd:
                after.append((t, h))
        return [_StagedTransfers(d.batch_isend_irecv(ops), after)]

    def _run(self, spec):
        for w in self._post(spec):
            w.wait()

    def exchange_halos(self):
        """new planes 1..d -> lower neighbour's n3+1..n3+d; new planes n3-d+1..n3 -> upper neighbour's 1-d..0
        (d = self.depth: 2, or 4 where two sweeps are taken per pass)"""
        for w in self._start_exchange():
            w.wait()

    def _start_exchange(self):
        if self.world == 1:
            return []
        spec = []
        d = self.depth
        # every rank lists its transfers lower-neighbour first: matching order on both sides of a link
        if self.lower is not None:
            spec.append((True, self.planes(1, d), self.lower))
            spec.append((False, self.planes(1 - d, d), self.lower))
        if self.upper is not None:
            spec.append((False, self.planes(self.n3 + 1, d), self.upper))
            spec.append((True, self.planes(self.n3 - d + 1, d), self.upper))
        return self._post(spec)

    def _exchange_periodic_ghosts(self):
        """grid_setup's periodic C faces (t_mg.f90:553-583): ghost plane 0 = plane n3g, ghost plane n3g+1 = plane 1,
        whole planes including the x/y ghosts the A and B faces filled before"""
        if self.world == 1:
            return
        bc = self.grid.bc6()
        first, last = 0, self.world - 1
        spec = []
        if bc[4] == lib.BC_PERIODIC:            # C0: plane 0 on the first rank <- plane n3g on the last
            if self.rank == last:
                spec.append((True, self.planes(self.n3, 1), first))
            if self.rank == first:
                spec.append((False, self.planes(0, 1), last))
        self._run(spec)
        spec = []
        if bc[5] == lib.BC_PERIODIC:            # C1: plane n3g+1 on the last rank <- plane 1 on the first
            if self.rank == first:
                spec.append((True, self.planes(1, 1), last))
            if self.rank == last:
                spec.append((False, self.planes(self.n3 + 1, 1), first))
        self._run(spec)

    # ---- the level operators -------------------------------------------------------------------
    def setup(self, init: bool):
        """grid_setup on every slab, then the planes only another rank can supply"""
        self.lv.setup(init)
        if self.periodic_z:
            self._exchange_periodic_ghosts()
        self.exchange_halos()
        self.lv.halo_written()

    def sweep(self, reduce: Optional[int] = None):
        """one red-black sweep of the whole level (gs_step, m_gs_CDS.F90:296-315).  The first and the last z chunk
        of the slab are swept first; their new planes travel to the neighbours while the rest of the slab is swept
        (the communication backend orders its transfers after the kernels already enqueued on the current stream,
        so the inner launch, enqueued afterwards, runs beside them)."""
        red = self.reduce if reduce is None else reduce
        # (the edge part of a single sweep completes the two planes a side a neighbour needs for its next single sweep;
        # where four are kept current for passes of two sweeps, planes 3, 4 and n3-3, n3-2 are ready only after the
        # whole sweep: such a level's odd single sweeps are exchanged after it)
        if self.world == 1 or not self.overlap or self.depth != 2:
            self.lv.sweep(1, red)
            self.exchange_halos()
            return
        self.lv.sweep_part(lib.PART_EDGE, red)
        works = self._start_exchange()
        self.lv.sweep_part(lib.PART_INNER, red)
        for w in works:
            w.wait()

    def sweep_pair(self, reduce: Optional[int] = None):
        """two red-black sweeps in one pass over the slab and ONE exchange (of four planes): the form gs_V's legs
        take (m_gs_CDS.F90:240-242, 271-273).  Needs self.depth == 4.  Reductions are those of the second sweep."""
        if self.depth != 4:
            raise RuntimeError("sweep_pair needs four halo planes (choose_depth)")
        red = self.reduce if reduce is None else reduce
        if self.world == 1 or not self.overlap:
            self.lv.sweep(2, red)
            self.exchange_halos()
            return
        self.lv.sweep_part(lib.PART_EDGE2, red)
        works = self._start_exchange()
        self.lv.sweep_part(lib.PART_INNER2, red)
        for w in works:
            w.wait()

    def result(self):
        """(max (vv-V)^2, sum|V|, exact sum in units of 2^-72) over the WHOLE level, identical on every rank"""
        if self.world == 1:
            return self.lv.result()
        return self.result_over_ranks()

    def result_over_ranks(self):
        """the all-reduced record.  Over RCCL the slab's record is reduced where the sweep left it on the device
        (mgx_level_result_device): no copy to the host and back before the collectives, one wait for the reduced
        numbers.  Max dV^2 travels as the bits of a double that is never negative: integer order is its order."""
        import torch
        if self._stream_ordered() and hasattr(self.lv, "result_device"):
            ptr = self.lv.result_device()
            rec = self.__dict__.get("_rec")
            if rec is None or rec[0] != ptr:
                t = torch.as_tensor(_DevicePlanes(ptr, 64), device=torch.device("cuda", self.lv.ctx.device)).view(torch.int64)
                rec = self._rec = (ptr, t)
            mx, dg = rec[1][:7].clone(), rec[1][1:5].clone()
            self.dist.all_reduce(mx, op=self.dist.ReduceOp.MAX, group=self.group)
            self.dist.all_reduce(dg, op=self.dist.ReduceOp.SUM, group=self.group)
            v = torch.cat([mx, dg]).tolist()                       # the one wait
            if v[6]:
                raise lib.MgxError(-2, f"a sweep of level {self.grid.layer} gave up inside the kernel on some rank: the field is not valid")
            total = sum(int(d) << (32 * i) for i, d in enumerate(v[7:11]))
            e2g = struct.unpack("<d", struct.pack("<q", v[0]))[0]
            return e2g, (float("nan") if v[5] else total / SUM_UNITS), total
        e2, s, units = self.lv.result()
        dev = self._scalar_device()
        bad = 1.0 if (s != s) else 0.0
        mx = torch.tensor([e2, bad], dtype=torch.float64, device=dev)
        self.dist.all_reduce(mx, op=self.dist.ReduceOp.MAX, group=self.group)
        digits = torch.tensor([(units >> (32 * i)) & 0xFFFFFFFF for i in range(4)], dtype=torch.int64, device=dev)
        self.dist.all_reduce(digits, op=self.dist.ReduceOp.SUM, group=self.group)
        total = sum(int(v) << (32 * i) for i, v in enumerate(digits.tolist()))
        e2g, badg = mx.tolist()
        return e2g, (float("nan") if badg else total / SUM_UNITS), total

    def download(self) -> np.ndarray:
        return self.lv.download()


# ----------------------------------------------------------------------------------------------------
# the whole hierarchy over z slabs: the reference's cycle control (cycle.Solver) with the fine levels decomposed

def aligned_splits(levels: List[G.MgGrid], world: int) -> List[List[Tuple[int, int]]]:
    """(z0, n3_local) per rank for every level, coarsest level split evenly and every finer slab the exact
    pre-image of its child slab, so that restriction and prolongation reach at most the halo planes: a level
    whose depth was halved doubles the slab, one whose depth was kept (t_mg.f90:191-194) keeps it."""
    out = [split_planes(int(levels[-1].n[2]), world)]
    for fine, coarse in zip(reversed(levels[:-1]), reversed(levels[1:])):
        nf, nc = int(fine.n[2]), int(coarse.n[2])
        if nf == 2 * nc:
            out.append([(2 * z0, 2 * k) for z0, k in out[-1]])
        elif nf == nc:
            out.append(list(out[-1]))
        else:
            raise ValueError(f"levels of depth {nf} and {nc}: z slabs need a depth that is halved exactly or kept")
    out.reverse()
    if world > 1 and min(k for _, k in out[-1]) < 2:
        raise ValueError(f"coarsest level of depth {int(levels[-1].n[2])} over {world} ranks: every slab needs at least 2 planes")
    return out


class SlabDevice(SlabLevel):
    """a slab with the interface cycle.Solver expects of a resident level (multigrid_amd.lib.DeviceLevel)"""

    def __init__(self, ctx, g: G.MgGrid, rank: int, world: int, z0: int, n3: int, group=None,
                 level_factory: Optional[Callable] = None, plane_tensor: Optional[Callable] = None, overlap: bool = True):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world, self.group = int(rank), int(world), group
        self.grid = g
        self.z0, self.n3 = int(z0), int(n3)
        self.n_global = tuple(int(v) for v in g.n)
        make = level_factory if level_factory is not None else ctx.level
        self.lv = make(g.layer, list(self.n_global), g.dtype, self.z0, self.n3)
        self._plane_tensor = plane_tensor if plane_tensor is not None else device_plane_tensor
        self.lower = rank - 1 if rank > 0 else None
        self.upper = rank + 1 if rank < world - 1 else None
        self.periodic_z = False
        self.reduce = lib.RED_ERR | lib.RED_SUM
        self.overlap = overlap and hasattr(self.lv, "sweep_part")
        self.depth = 2
        self.allow_pairs = True

    # parameters come through cycle.Solver.push_params
    def set_params(self, sor, a, bc):
        self.lv.set_params(sor, a, bc)
        self.periodic_z = bc[4] == lib.BC_PERIODIC or bc[5] == lib.BC_PERIODIC

    def set_boxes(self, *table):
        self.lv.set_boxes(*table)
        self.choose_depth(self.allow_pairs)      # (the library's answer depends on the box table: at most 32 boxes)

    def bring_back(self):
        self.lv.bring_back()

    def release(self):
        self.lv.release()

    def close(self):
        self.lv.close()

    def count_nonconst(self) -> int:
        return self.lv.count_nonconst()          # the library counts over the global level

    def refresh_halos(self):
        """after anything that rewrote the slab as a whole (set-up, restriction, prolongation)"""
        if self.periodic_z:
            self._exchange_periodic_ghosts()
        self.exchange_halos()
        self.lv.halo_written()

    def setup(self, init: bool):
        self.lv.setup(init)
        self.refresh_halos()

    def sweep(self, nsweeps: int = 1, reduce: Optional[int] = None):     # `nsweeps` x gs_step
        red = self.reduce if reduce is None else reduce
        s, nsweeps = 0, int(nsweeps)
        while self.depth == 4 and s + 2 <= nsweeps:                     # pairs of sweeps in one pass and one exchange
            SlabLevel.sweep_pair(self, red if s + 2 == nsweeps else 0)
            s += 2
        while s < nsweeps:
            SlabLevel.sweep(self, red if s == nsweeps - 1 else 0)
            s += 1

    def abs_sum(self) -> float:
        """grid_sum over the whole level: the slabs' exact integer sums added exactly"""
        units = self.lv.abs_sum_limbs()
        if self.world > 1:
            import torch
            dev = self._scalar_device()
            digits = torch.tensor([(units >> (32 * i)) & 0xFFFFFFFF for i in range(4)], dtype=torch.int64, device=dev)
            self.dist.all_reduce(digits, op=self.dist.ReduceOp.SUM, group=self.group)
            units = sum(int(v) << (32 * i) for i, v in enumerate(digits.tolist()))
        return units / SUM_UNITS


class ReplicatedDevice:
    """a level too thin to be decomposed: every rank holds it whole and does the same work on it.  This is the
    coarse-level gather of a decomposed solve done as an all-gather: each GPU plays the part of the one that would hold
    the gathered level (identical kernels, identical results, no communication while the cycle stays on such levels),
    and nothing has to be scattered on the way back up."""

    def __init__(self, ctx, g: G.MgGrid, level_factory: Optional[Callable] = None):
        n = [int(v) for v in g.n]
        make = level_factory if level_factory is not None else ctx.level
        self.lv = make(g.layer, n, g.dtype, 0, n[2])
        self.grid, self.z0, self.n3 = g, 0, n[2]

    def set_params(self, sor, a, bc):
        self.lv.set_params(sor, a, bc)

    def set_boxes(self, *table):
        self.lv.set_boxes(*table)

    def bring_back(self):
        self.lv.bring_back()

    def release(self):
        self.lv.release()

    def close(self):
        self.lv.close()

    def count_nonconst(self) -> int:
        return self.lv.count_nonconst()

    def setup(self, init: bool):
        self.lv.setup(init)

    def refresh_halos(self):
        pass

    def sweep(self, nsweeps: int = 1, reduce: int = lib.RED_ERR | lib.RED_SUM):
        self.lv.sweep(int(nsweeps), reduce)

    def result(self):
        return self.lv.result()

    def abs_sum(self) -> float:
        return self.lv.abs_sum_limbs() / SUM_UNITS

    def download(self) -> np.ndarray:
        return self.lv.download()


class SlabSolver:
    """the reference's cycle control (cycle.Solver) over z slabs: fine levels are SlabDevices, aligned so that
    restriction and prolongation are local given current halo planes and are followed by the halo refresh of the
    level they wrote; levels whose slabs would be thinner than `replicate_below` planes are ReplicatedDevices.  Between
    the last decomposed level and the first replicated one sits a slab of the replicated level's geometry: restriction
    fills it and the ranks all-gather their planes into their whole copies; prolongation reads it after every rank
    has copied its planes (and halo planes) out of its own whole copy."""

    @staticmethod
    def build(top: G.MgGrid, ctx, rank: int, world: int, group=None, level_factory=None, plane_tensor=None,
              transfer=None, overlap: bool = True, replicate_below: int = 16, pairs: bool = True, **kw):
        import torch.distributed as dist
        from . import cycle
        levels = list(top.levels())
        first_rep = len(levels)                  # index of the first replicated level
        if world > 1:
            for i, g in enumerate(levels):
                if int(g.n[2]) // world < max(2, int(replicate_below)):
                    first_rep = i
                    break
            while 0 < first_rep < len(levels) and int(levels[first_rep].n[2]) < 2 * world:
                first_rep -= 1                   # the slab between the two kinds of level needs two planes per rank
            if first_rep == 0:
                raise ValueError(f"top level of depth {int(top.n[2])} over {world} ranks: slabs would be thinner than "
                                 f"{max(2, int(replicate_below))} planes")
        sliced = levels[:first_rep + 1] if first_rep < len(levels) else levels
        splits = aligned_splits(sliced, world)
        by_layer = {g.layer: sp[rank] for g, sp in zip(sliced, splits)}
        rep_layers = {g.layer for g in levels[first_rep:]}
        edge_split = splits[-1] if first_rep < len(levels) else None
        do_restrict, do_prolong = transfer if transfer is not None else (lib.restrict, lib.prolong)
        planes_of = plane_tensor if plane_tensor is not None else device_plane_tensor

        def slab_of(g):
            z0, n3 = by_layer[g.layer]
            d = SlabDevice(ctx, g, rank, world, z0, n3, group=group, level_factory=level_factory,
                           plane_tensor=plane_tensor, overlap=overlap)
            d.allow_pairs = pairs                # two sweeps per pass and exchange where the library offers them
            return d

        class _Solver(cycle.Solver):
            edge = None                          # the slab of the first replicated level's geometry
            reference_order_available = world == 1 and level_factory is None   # a slab has no kernel for that order

            def make_level(self, g):
                if g.layer in rep_layers:
                    return ReplicatedDevice(ctx, g, level_factory)
                return slab_of(g)

            def edge_slab(self, g):
                if self.edge is None:
                    self.edge = slab_of(g)
                    self.edge.set_params(g.sor, [float(v) for v in g.a], g.bc6())
                    self.edge.set_boxes(*G.box_table(g))
                return self.edge

            def host_wait(self, dev):
                if not dev._stream_ordered():    # gloo reads device planes from the host
                    dev.lv.ctx.sync()
                    import torch
                    torch.cuda.synchronize()

            def restriction(self, g):
                c = g.child
                if g.layer in rep_layers:        # both whole: every rank does the same
                    do_restrict(g.dev.lv, c.dev.lv, g.RES_method)
                elif c.layer not in rep_layers:
                    do_restrict(g.dev.lv, c.dev.lv, g.RES_method)
                    c.dev.refresh_halos()
                else:
                    t = self.edge_slab(c)
                    t.bring_back()
                    do_restrict(g.dev.lv, t.lv, g.RES_method)
                    for r, (z0r, kr) in enumerate(edge_split):      # all-gather of the owned planes
                        dst = planes_of(c.dev.lv, z0r + 1, kr)
                        if r == rank:
                            dst.copy_(planes_of(t.lv, 1, kr))
                        src = r if group is None else dist.get_global_rank(group, r)
                        if t._stream_ordered():
                            dist.broadcast(dst, src=src, group=group)
                        else:                    # gloo moves host memory: stage the device planes
                            self.host_wait(t)
                            h = dst.cpu()
                            dist.broadcast(h, src=src, group=group)
                            if r != rank:
                                dst.copy_(h)
                    self.host_wait(t)
                    c.dev.setup(False)           # grid_setup(child), the end of grid_restriction (t_mg_interp.f90:9-21)

            def prolongation(self, g):
                p = g.parent
                if p.layer in rep_layers:
                    do_prolong(g.dev.lv, p.dev.lv, g.PRO_method)
                elif g.layer not in rep_layers:
                    do_prolong(g.dev.lv, p.dev.lv, g.PRO_method)
                    p.dev.refresh_halos()
                else:
                    t = self.edge_slab(g)
                    t.bring_back()
                    lo, hi = max(-1, -t.z0), min(t.n3 + 2, int(g.n[2]) + 1 - t.z0)   # local planes that exist globally
                    self.host_wait(t)
                    planes_of(t.lv, lo, hi - lo + 1).copy_(planes_of(g.dev.lv, t.z0 + lo, hi - lo + 1))
                    self.host_wait(t)
                    t.lv.halo_written()
                    do_prolong(t.lv, p.dev.lv, g.PRO_method)
                    p.dev.refresh_halos()

            def close(self):
                if self.edge is not None:
                    self.edge.close()
                    self.edge = None
                super().close()

        return _Solver(top, ctx=ctx if ctx is not None else _NoContext(), **kw)


class _NoContext:
    """stands in for lib.Context when the levels come from a factory (tests)"""

    def close(self):
        pass
