"""TEST INFRASTRUCTURE: a slab level that lives in numpy and is swept by the CPU oracle, with the interface
multigrid_amd.slab.SlabLevel expects of a level.  It lets the halo exchange and the all-reduced reductions run on
CPU ranks (gloo) and be checked against the oracle's sweep of the undivided level."""
import numpy as np
import torch

from oracle import oracle as orc


class FakeSlabLevel:
    def __init__(self, layer, n, dtype, z0, n3_local, global_init=None):
        self.n_global = tuple(int(v) for v in n)
        self.dtype = np.dtype(dtype)
        self.z0, self.n3 = int(z0), int(n3_local)
        n1, n2, _ = self.n_global
        self.V = np.zeros((self.n3 + 4, n2 + 2, n1 + 2), dtype=self.dtype)     # local planes -1 .. n3+2
        self.global_init = global_init
        self.res = (0.0, 0.0, 0)

    def set_params(self, sor, a, bc):
        self.sor, self.a, self.bc = float(sor), tuple(a), tuple(bc)

    def set_boxes(self, place, val, rho, constant):
        self.boxes = [orc.Box(tuple(int(v) for v in p), float(v), float(r), bool(c))
                      for p, v, r, c in zip(np.asarray(place).reshape(-1, 6), val, rho, constant)]

    def _global_level(self):
        return orc.Level(n=self.n_global, bc=self.bc, sor=self.sor, a=self.a, boxes=self.boxes)

    def _pseudo_level(self):
        """local planes 0 .. n3+1 as the interior of a level of depth n3+2 (boxes shifted in z)"""
        sh = 1 - self.z0
        bx = [orc.Box(tuple(b.place[:4]) + (b.place[4] + sh, b.place[5] + sh) if any(b.place) else b.place, b.val, b.rho,
                      b.constant) for b in self.boxes]
        n1, n2, _ = self.n_global
        return orc.Level(n=(n1, n2, self.n3 + 2), bc=self.bc, sor=self.sor, a=self.a, boxes=bx)

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
                self.V[lz + 1] = Vg[gz]

    def halo_written(self):
        pass

    def sweep(self, nsweeps, reduce):
        assert nsweeps == 1
        n3 = self.n3
        keep = {lz: self.V[lz + 1].copy() for lz in (-1, 0, n3 + 1, n3 + 2)}
        old = self.V[2:n3 + 2, 1:-1, 1:-1].copy()
        P = self._pseudo_level()
        flip = (self.z0 + 1) & 1
        orc.half_sweep(P, self.V, 1 ^ flip)          # black = odd x + y + z (global)
        # black never runs on a global ghost plane: undo it there before red looks at it
        if self.z0 == 0:
            self.V[1] = keep[0]
        if self.z0 + n3 == self.n_global[2]:
            self.V[n3 + 2] = keep[n3 + 1]
        orc.half_sweep(P, self.V, 0 ^ flip)
        for lz, v in keep.items():
            self.V[lz + 1] = v
        new = self.V[2:n3 + 2, 1:-1, 1:-1]
        d = new - old
        e2 = float((d * d).max()) if (reduce & 1) else 0.0
        s, units = (0.0, 0)
        if reduce & 2:
            n1, n2, _ = self.n_global
            s, units = orc.abs_sum_exact(orc.Level(n=(n1, n2, n3)), np.ascontiguousarray(self.V[1:n3 + 3]))
        self.res = (e2, s, units)

    def sweep_part(self, part, reduce):
        """edge part = the whole sweep (the new boundary planes exist afterwards); inner part = nothing left to do"""
        if part == 1:
            self.sweep(1, reduce)

    def result(self):
        return self.res

    def download(self):
        return self.V[1:self.n3 + 3].copy()


def plane_tensor(lv: FakeSlabLevel, z: int, count: int):
    return torch.from_numpy(lv.V[z + 1:z + 1 + count])


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
        return orc.abs_sum_exact(orc.Level(n=(n1, n2, self.n3)), np.ascontiguousarray(self.V[1:self.n3 + 3]))[1]

    def gather_global(self):
        """the undivided level as every rank sees it: owned planes of all ranks + the two global ghost planes"""
        import torch.distributed as dist
        n1, n2, n3g = self.n_global
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, (self.z0, self.n3, self.V[1:self.n3 + 3].copy()))
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
                self.V[lz + 1] = Vg[gz]

    def setup(self, init):
        Vg = self.gather_global()
        orc.setup(self._global_level(), Vg, bool(init))
        self.take_from_global(Vg)

    def sweep(self, nsweeps, reduce):
        for _ in range(int(nsweeps)):
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
