"""Multi-rank cycle control on CPU: multigrid_amd.slab.SlabSolver (the reference's m_gs_CDS over z slabs) on gloo
ranks with oracle-backed fake levels, against the oracle's red-black solve of the undivided hierarchy: residual
history, sweeps per level and the top-level potential must agree bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # x and y are too small to be halved, z is: three levels of depth 160, 80, 40 (t_mg.f90:183-201)
    "bottom_up_mixed": dict(dtype=np.float64, text="""
cell 1.6 1.6 16.
max-layers 3
boxes 2
method bottom-up
begin layers
  sor 1.5
  tol 1.e-4
  mesh-size 16 16 160
  bc a periodic
  bc b neumann
  bc c- dirichlet
  bc c+ neumann
end layers
begin box
  llc 0.4 0.4 0.
  cell 0.8 0.8 0.4
  value 0.5
end box
begin box
  llc 0.4 0.4 15.2
  cell 0.8 0.8 0.4
  value -0.5
end box
"""),
    "v_cycle_periodic_z_half": dict(dtype=np.float32, text="""
cell 1.6 1.2 16.
max-layers 3
boxes 2
method v-cycle
begin layers
  sor 1.5
  tol 1.e-3
  v-steps 2
  mesh-size 16 12 160
  bc all periodic
  interp half
end layers
begin box
  llc 0.4 0.3 3.
  cell 0.8 0.6 0.8
  value 0.5
end box
begin box
  llc 0.4 0.3 11.
  cell 0.8 0.6 1.6
  value -0.5
  rho 2.
  constant false
end box
"""),
    # z is the direction that is kept: every level has depth 16
    "v_cycle_kept_z": dict(dtype=np.float64, text="""
cell 12. 1.6 1.6
max-layers 3
boxes 1
method v-cycle
begin layers
  sor 1.4
  tol 1.e-3
  v-steps 1
  mesh-size 120 16 16
  bc all neumann
  bc c periodic
end layers
begin box
  llc 5. 0.4 0.4
  cell 2. 0.8 0.8
  value 1.
end box
"""),
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _read(case):
    from multigrid_amd import mgio
    IO = mgio.MgIO(text=case["text"])
    return mgio.iomg_read(IO, dtype=case["dtype"]), mgio.read_method(IO)


def _worker(rank, world, port, name, outdir, replicate_below=4, pairs=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multigrid_amd import slab
        from tests import slab_fake
        from tests.helpers import oracle_levels
        from oracle import oracle as orc
        case = CASES[name]
        top, method = _read(case)
        V, hist, per = orc.solve(oracle_levels(top), method, case["dtype"], order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
        keep = hist
        extra = {}
        if pairs:                 # the levels offer two sweeps per pass: legs of a cycle run as passes with ONE exchange
            slab_fake.FakeSlabLevel.pairs = True
            extra = dict(print_sums=False)         # (with the print-only sums every sweep of a leg is a call of its own)
            keep = hist[hist["kind"] != 1]
        s = slab.SlabSolver.build(top, None, rank, world, level_factory=slab_fake.FakeCycleLevel,
                                  plane_tensor=slab_fake.plane_tensor,
                                  transfer=(slab_fake.fake_restrict, slab_fake.fake_prolong),
                                  replicate_below=replicate_below, **extra)
        kinds = [type(g.dev).__name__ for g in top.levels()]
        open(os.path.join(outdir, f"kinds{rank}"), "w").write(",".join(kinds))
        depths = [getattr(g.dev, "depth", 0) for g in top.levels()]
        open(os.path.join(outdir, f"depths{rank}"), "w").write(",".join(map(str, depths)))
        s.mg_gs_cds(method)
        h = s.history_array()
        assert [g.itt for g in top.levels()] == per, ([g.itt for g in top.levels()], per)
        assert len(h) == len(keep)
        for f in ("layer", "kind", "itt", "tol", "sum", "err"):
            assert np.array_equal(h[f], keep[f]), f
        d = top.dev
        mine = d.download()
        assert np.array_equal(mine[1:-1].view(np.uint8), V[d.z0 + 1:d.z0 + d.n3 + 1].view(np.uint8))
        open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(CASES))
def test_slab_cycle_matches_undivided_solve(name, world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("name,world,below,kinds", [
    ("bottom_up_mixed", 2, 30, "SlabDevice,SlabDevice,ReplicatedDevice"),
    ("bottom_up_mixed", 3, 30, "SlabDevice,ReplicatedDevice,ReplicatedDevice"),
    ("v_cycle_periodic_z_half", 2, 50, "SlabDevice,ReplicatedDevice,ReplicatedDevice"),
    ("v_cycle_periodic_z_half", 3, 20, "SlabDevice,SlabDevice,ReplicatedDevice"),
])
def test_slab_cycle_with_replicated_coarse_levels(name, world, below, kinds, tmp_path):
    """levels whose slabs would be thinner than `below` planes live whole on every rank (the coarse-level gather):
    same history, counts and potential as the undivided solve"""
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path), below), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
    assert (tmp_path / "kinds0").read_text() == kinds


@pytest.mark.parametrize("name,world,below", [("v_cycle_periodic_z_half", 2, 4), ("v_cycle_periodic_z_half", 3, 20),
                                              ("bottom_up_mixed", 2, 4)])
def test_slab_cycle_with_two_sweeps_per_exchange(name, world, below, tmp_path):
    """levels that offer passes of two sweeps (four halo planes, one exchange per pass): the legs of a V cycle take
    them, single sweeps (grid_solve's loop, odd counts) go on with four-plane exchanges; same history, counts and
    potential as the undivided solve"""
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path), below, True), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
    assert "4" in (tmp_path / "depths0").read_text()


def test_aligned_splits():
    from multigrid_amd import slab
    top, _ = _read(CASES["bottom_up_mixed"])
    assert [int(g.n[2]) for g in top.levels()] == [160, 80, 40]
    sp = slab.aligned_splits(list(top.levels()), 3)
    assert sp[-1] == [(0, 14), (14, 13), (27, 13)] and sp[0] == [(0, 56), (56, 52), (108, 52)]
    with pytest.raises(ValueError):
        slab.aligned_splits(list(top.levels()), 21)
    top, _ = _read(CASES["v_cycle_kept_z"])
    assert [tuple(int(v) for v in g.n) for g in top.levels()] == [(120, 16, 16), (60, 16, 16), (30, 16, 16)]
    assert slab.aligned_splits(list(top.levels()), 2) == [[(0, 8), (8, 8)]] * 3
