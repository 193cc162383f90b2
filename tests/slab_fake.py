"""TEST INFRASTRUCTURE: a slab level that lives in numpy and is swept by the CPU oracle, with the interface
multigrid_amd.slab.SlabLevel expects of a level.  It lets the halo exchange and the all-reduced reductions run on
CPU ranks (gloo) and be checked against the oracle's sweep of the undivided level."""
import numpy as np
import torch

from oracle import oracle as orc


OFF = 3      # V[lz + OFF] = local plane lz; local planes -3 .. n3+4 exist (four halo planes per side, like a device slab)


class FakeSlabLevel:
    pairs = False        # answer of pair_pass(): set True to have the slab layer take two sweeps per exchange

    def __init__(self, layer, n, dtype, z0, n3_local, global_init=None):
        self.n_global = tuple(int(v) for v in n)
        self.dtype = np.dtype(dtype)
        self.z0, self.n3 = int(z0), int(n3_local)
        n1, n2, _ = self.n_global
        self.V = np.zeros((self.n3 + 8, n2 + 2, n1 + 2), dtype=self.dtype)     # local planes -3 .. n3+4
        self.global_init = global_init
        self.res = (0.0, 0.0, 0)

    def pair_pass(self):
        return bool(self.pairs)

    def set_params(self, sor, a, bc):
        self.sor, self.a, self.bc = float(sor), tuple(a), tuple(bc)

    def set_boxes(self, place, val, rho, constant):
        self.boxes = [orc.Box(tuple(int(v) for v in p), float(v), float(r), bool(c))
                      for p, v, r, c in zip(np.asarray(place).reshape(-1, 6), val, rho, constant)]

    def _global_level(self):
        return orc.Level(n=self.n_global, bc=self.bc, sor=self.sor, a=self.a, boxes=self.boxes)

    def _pseudo_level(self, grow):
        """local planes 1-grow .. n3+grow as the interior of a level of depth n3 + 2 grow (boxes shifted in z)"""
        sh = grow - self.z0
        bx = [orc.Box(tuple(b.place[:4]) + (b.place[4] + sh, b.place[5] + sh) if any(b.place) else b.place, b.val, b.rho,
                      b.constant) for b in self.boxes]
        n1, n2, _ = self.n_global
        return orc.Level(n=(n1, n2, self.n3 + 2 * grow), bc=self.bc, sor=self.sor, a=self.a, boxes=bx)

    def setup(self, init):
        G = self._global_level()
        n1, n2, n3g = self.n_global
        if init or self.global_init is None:
            Vg = np.zeros((n3g + 2, n2 + 2, n1 + 2), dtype=self.dtype)
        else:
            Vg = np.array(self.global_init, dtype=self.dtype)
        orc.setup(G, Vg, bool(init))
        whole = self.z0 == 0 and self.n3 == n3g
        self.V[:] = 0
        for lz in range(-1, self.n3 + 3):
            gz = lz + self.z0
            inner_halo = (lz <= 0 and self.z0 > 0) or (lz >= self.n3 + 1 and self.z0 + self.n3 < n3g)
            periodic_ghost = ((gz == 0 and self.bc[4] == 1) or (gz == n3g + 1 and self.bc[5] == 1)) and not whole
            if 0 <= gz <= n3g + 1 and not inner_halo and not periodic_ghost:    # the rest is the exchange's job
                self.V[lz + OFF] = Vg[gz]

    def halo_written(self):
        pass

    def _half_sweeps(self, count, reduce):
        """`count` colour half-sweeps (black first) over the slab grown by what the later ones still need, as the
        device kernels do: half-sweep s of `count` updates local planes 1-(count-1-s) .. n3+(count-1-s) (never a global
        ghost plane); what it computes beyond is never read by a plane that counts.  Halo planes are put back at the
        end; the reductions are those of the last full sweep over the slab's own planes."""
        n3 = self.n3
        grow = count - 1                                    # planes the first half-sweep reaches beyond the slab
        view = self.V[OFF - grow:OFF + n3 + grow + 2]          # planes -grow .. n3+grow+1: the pseudo level with its z ghosts
        halo = [lz for lz in range(-3, n3 + 5) if lz < 1 or lz > n3]
        keep = {lz: self.V[lz + OFF].copy() for lz in halo}
        P = self._pseudo_level(grow)
        flip = (grow - self.z0) & 1                          # parity of pseudo z against global z
        n3g = self.n_global[2]
        old = None
        for s in range(count):
            if s == count - 2:
                old = self.V[OFF + 1:OFF + n3 + 1, 1:-1, 1:-1].copy()
            colour = 1 if s % 2 == 0 else 0                  # black = odd x + y + z (global) first
            orc.half_sweep(P, view, colour ^ flip)
            # a half-sweep never runs on a global ghost plane (or beyond): undo it there before the next one looks
            for lz in halo:
                gz = lz + self.z0
                if gz < 1 or gz > n3g:
                    self.V[lz + OFF] = keep[lz]
        for lz, v in keep.items():
            self.V[lz + OFF] = v
        new = self.V[OFF + 1:OFF + n3 + 1, 1:-1, 1:-1]
        d = new - old
        e2 = float((d * d).max()) if (reduce & 1) else 0.0
        s_, units = (0.0, 0)
        if reduce & 2:
            n1, n2, _ = self.n_global
            s_, units = orc.abs_sum_exact(orc.Level(n=(n1, n2, n3)), np.ascontiguousarray(self.V[OFF:OFF + n3 + 2]))
        self.res = (e2, s_, units)

    def sweep(self, nsweeps, reduce):
        assert nsweeps in (1, 2)
        self._half_sweeps(2 * int(nsweeps), reduce)

    def sweep_part(self, part, reduce):
        """edge part = the whole pass, of which only the planes the library's edge launch completes are shown (two a
        side of a single sweep, four of a pass of two): the others are NaN until the inner part, so that a host which
        sends planes before they exist spoils the result"""
        if part in (1, 3):
            self.sweep(1 if part == 1 else 2, reduce)
            keep = 2 if part == 1 else 4
            lo, hi = OFF + keep + 1, OFF + self.n3 - keep + 1         # planes keep+1 .. n3-keep
            self._due = (part + 1, lo, hi, self.V[lo:hi].copy()) if hi > lo else (part + 1, 0, 0, None)
            if hi > lo:
                self.V[lo:hi] = np.nan
        else:
            want, lo, hi, planes = self._due
            assert part == want, "inner part without its edge part"
            if planes is not None:
                self.V[lo:hi] = planes
            self._due = None

    def result(self):
        return self.res

    def download(self):
        return self.V[OFF:OFF + self.n3 + 2].copy()


def plane_tensor(lv: FakeSlabLevel, z: int, count: int):
    return torch.from_numpy(lv.V[z + OFF:z + OFF + count])


class FakeCycleLevel(FakeSlabLevel):
    """the fake slab with the rest of the interface multigrid_amd.slab.SlabDevice forwards to.  Operators that look
    at the level as a whole (set-up of existing values, restriction, prolongation) gather the slabs of all ranks,
    apply the oracle to the undivided level and keep this rank's planes; planes that only the halo exchange may
    supply are left stale on purpose, so a missing exchange shows up as a wrong result."""

    def bring_back(self):
        self.V[:] = 0

    def release(self):
        pass

    def close(self):
        pass

    def count_nonconst(self):
        return orc.count_nonconst(self._global_level())

    def abs_sum_limbs(self):
        n1, n2, _ = self.n_global
        return orc.abs_sum_exact(orc.Level(n=(n1, n2, self.n3)), np.ascontiguousarray(self.V[OFF:OFF + self.n3 + 2]))[1]

    def gather_global(self):
        """the undivided level as every rank sees it: owned planes of all ranks + the two global ghost planes"""
        import torch.distributed as dist
        n1, n2, n3g = self.n_global
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, (self.z0, self.n3, self.V[OFF:OFF + self.n3 + 2].copy()))
        Vg = np.zeros((n3g + 2, n2 + 2, n1 + 2), dtype=self.dtype)
        for z0, n3, pl in parts:
            Vg[z0 + 1:z0 + n3 + 1] = pl[1:-1]
            if z0 == 0:
                Vg[0] = pl[0]
            if z0 + n3 == n3g:
                Vg[n3g + 1] = pl[-1]
        return Vg

    def take_from_global(self, Vg):
        """owned planes and global ghost planes; inner halos and periodic ghosts belong to the exchange"""
        n3g = self.n_global[2]
        whole = self.z0 == 0 and self.n3 == n3g
        for lz in range(0, self.n3 + 2):
            gz = lz + self.z0
            inner_halo = (lz == 0 and self.z0 > 0) or (lz == self.n3 + 1 and self.z0 + self.n3 < n3g)
            periodic_ghost = ((gz == 0 and self.bc[4] == 1) or (gz == n3g + 1 and self.bc[5] == 1)) and not whole
            if not inner_halo and not periodic_ghost:
                self.V[lz + OFF] = Vg[gz]

    def setup(self, init):
        Vg = self.gather_global()
        orc.setup(self._global_level(), Vg, bool(init))
        self.take_from_global(Vg)

    def sweep(self, nsweeps, reduce):
        nsweeps = int(nsweeps)
        if nsweeps == 2 and self.pairs:
            FakeSlabLevel.sweep(self, 2, reduce)
            return
        for _ in range(nsweeps):
            FakeSlabLevel.sweep(self, 1, reduce)


def fake_restrict(parent: FakeCycleLevel, child: FakeCycleLevel, method: int):
    P, Cl = parent._global_level(), child._global_level()
    P.res_method = int(method)
    Vp, Vc = parent.gather_global(), child.gather_global()
    orc.restrict(P, Vp, Cl, Vc)
    child.take_from_global(Vc)


def fake_prolong(child: FakeCycleLevel, parent: FakeCycleLevel, method: int):
    P, Cl = parent._global_level(), child._global_level()
    Cl.pro_method = int(method)
    Vp, Vc = parent.gather_global(), child.gather_global()
    orc.prolong(Cl, Vc, P, Vp)
    parent.take_from_global(Vp)
