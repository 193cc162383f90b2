"""libmgx_rccl.so (include/mgx_rccl.h): halo exchange and exact reductions of z slabs over RCCL inside the library.
CPU: the library loads and exports what its header declares.  GPU (one card): a communicator of one rank — the
reductions go through ncclAllReduce and must return what mgx_level_result holds; the periodic ghost planes travel
through a grouped ncclSend/ncclRecv of the rank to itself and must equal a plain copy.  More than one rank needs more
than one GPU, which the test machines do not have: multi-rank exchanges are tested through multigrid_amd/slab.py."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    text = open(os.path.join(ROOT, "include", "mgx_rccl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", text)))


def test_companion_library_exports_its_header():
    from multigrid_amd import rccl
    assert os.path.exists(rccl.RCCL_LIB_PATH), "build it first: make -C multigrid_amd/csrc"
    assert sorted(rccl.SYMBOLS) == declared()
    try:
        dll = ctypes.CDLL(rccl.RCCL_LIB_PATH)
    except OSError as e:
        if "librccl" in str(e):
            pytest.skip(f"librccl.so does not load on this machine: {e}")
        raise
    assert not [n for n in declared() if not hasattr(dll, n)]


@pytest.mark.gpu
def test_single_rank_communicator_reduces_and_wraps():
    import torch   # noqa: F401  before the HIP library: one HIP runtime per process
    from multigrid_amd import lib, rccl
    ctx = lib.Context(0)
    n = (48, 40, 32)
    lv = ctx.level(1, n, np.float64, 0, n[2])
    comm = None
    try:
        lv.set_params(1.6, (0.3, 0.3, 0.4), (1, 1, 1, 1, 1, 1))
        lv.set_boxes([(10, 20, 10, 20, 5, 9), (30, 40, 10, 30, 20, 28)], [0.5, -0.5], [1.0, 1.0], [1, 1])
        lv.bring_back()
        lv.setup(True)
        (comm,) = rccl.Comm.create_all([ctx])
        for _ in range(3):
            lv.sweep(1, lib.RED_ERR | lib.RED_SUM)
            comm.exchange(lv)                      # one rank: no planes travel; the streams still meet
        for planes in (2, 4):                      # a split sweep with the exchange around its inner part
            lv.sweep_part(lib.PART_EDGE, lib.RED_ERR | lib.RED_SUM)
            comm.exchange_begin(lv, planes)
            lv.sweep_part(lib.PART_INNER, lib.RED_ERR | lib.RED_SUM)
            comm.exchange_end()
        with pytest.raises(lib.MgxError):
            comm.exchange_begin(lv, 3)
        want = lv.result()
        assert comm.allreduce_result(lv) == want and want[1] > 0
        V = lv.download()                          # (n3+2 [+ halo planes], n2+2, n1+2) ghosted
        comm.periodic_ghosts(lv, True, True)
        ctx.sync()
        W = lv.download()
        assert np.array_equal(W[0], V[n[2]]) and np.array_equal(W[n[2] + 1], V[1])
        assert np.array_equal(W[1:n[2] + 1], V[1:n[2] + 1])
    finally:
        if comm is not None:
            comm.close()
        lv.close()
        ctx.close()


@pytest.mark.gpu
def test_fortran_slab_driver_over_the_library_exchange():
    """multigrid_amd/fortran/slab_rccl (Fortran, one process, mgx_comm_create_all + the *_all calls, split sweeps with
    the exchange in between) on the GPUs present — one here — against the same sweeps of the undivided level through
    the Python binding: same max dV^2, same exact sum."""
    import subprocess
    import torch   # noqa: F401
    from multigrid_amd import lib
    exe = os.path.join(ROOT, "multigrid_amd", "fortran", "slab_rccl")
    assert os.path.exists(exe), "build it first: make -C multigrid_amd/fortran"
    n, sweeps = 64, 11
    got = []
    for pairs in ("0", "1"):       # "1": two sweeps per pass and per exchange of four planes, the odd last one alone
        r = subprocess.run([exe, "1", str(n), str(sweeps), pairs], capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        line = [l for l in r.stdout.splitlines() if "err_sq_max" in l][0].split()
        hexes = [l for l in r.stdout.splitlines() if "limbs" in l][0].split()
        got.append((float(line[2]), float(line[4]), (int(hexes[2], 16) << 64) | int(hexes[3], 16)))
    assert got[0] == got[1]
    e2, s, units = got[0]
    ctx = lib.Context(0)
    lv = ctx.level(1, (n, n, n), np.float64)
    try:
        lv.set_params(1.6, (1 / 3,) * 3, (2, 2, 2, 2, 1, 1))
        q = n // 4
        lv.set_boxes([(q, 3 * q, q, 3 * q, n // 8, n // 8 + 1), (q, 3 * q, q, 3 * q, 7 * n // 8, 7 * n // 8 + 1)],
                     [0.5, -0.5], [1.0, 1.0], [1, 1])
        lv.bring_back()
        lv.setup(True)
        lv.sweep(sweeps, lib.RED_ERR | lib.RED_SUM)
        we2, ws, wunits = lv.result()
    finally:
        lv.close()
        ctx.close()
    assert units == wunits and s == ws and e2 == we2 and ws > 0


@pytest.mark.gpu
def test_python_slab_reduction_stays_on_the_device_over_rccl():
    """multigrid_amd/slab.py over the RCCL backend (a process group of one rank: the collectives run, over one card):
    the slab's record is all-reduced from where the sweep left it and must decode to what mgx_level_result holds"""
    import socket
    import torch
    import torch.distributed as dist
    from multigrid_amd import grid as G, lib, mgio, slab
    from tests.helpers import load
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    ctx = None
    try:
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        ctx = lib.Context(0, stream.cuda_stream)
        top = mgio.iomg_read(mgio.MgIO(text=str(load("solve_64_bu_f64")["input"])), dtype=np.float64)
        sl = slab.SlabLevel(ctx, top, 0, 1)
        sl.lv.bring_back()
        sl.lv.setup(True)
        for k in range(3):
            sl.lv.sweep(1, lib.RED_ERR | lib.RED_SUM)
            want = sl.lv.result()
            got = sl.result_over_ranks()
            assert got == want and want[2] > 0, (k, got, want)
        assert sl._rec[1].is_cuda
    finally:
        if ctx is not None:
            ctx.close()
        torch.cuda.set_stream(torch.cuda.default_stream())
        dist.destroy_process_group()
