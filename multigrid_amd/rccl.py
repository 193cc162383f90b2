"""ctypes binding of libmgx_rccl.so (include/mgx_rccl.h): halo exchange and exact reductions of z-slab levels over
RCCL inside the library, for hosts without PyTorch.  The Python host itself exchanges through torch.distributed
(multigrid_amd/slab.py); this binding exists for tests and for Python hosts that want RCCL without torch.

There is no fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

from . import lib

RCCL_LIB_PATH = os.environ.get("MGX_RCCL_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmgx_rccl.so"))
COMM_ID_BYTES = 128

SYMBOLS = [
    "mgx_comm_last_error", "mgx_comm_unique_id", "mgx_comm_create", "mgx_comm_create_all", "mgx_comm_destroy",
    "mgx_slab_exchange", "mgx_slab_exchange_all", "mgx_slab_exchange_begin", "mgx_slab_exchange_end",
    "mgx_slab_exchange_begin_all", "mgx_slab_exchange_end_all", "mgx_slab_periodic_ghosts", "mgx_slab_periodic_ghosts_all",
    "mgx_allreduce_result", "mgx_allreduce_result_all",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib.load()                                    # libmgx.so first: the companion links against it
    if not os.path.exists(RCCL_LIB_PATH):
        raise ImportError(f"{RCCL_LIB_PATH} not found: build it with `make -C multigrid_amd/csrc`")
    L = C.CDLL(RCCL_LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, pvp = C.c_void_p, C.POINTER(C.c_void_p)
    pdbl, pu64 = C.POINTER(C.c_double), C.POINTER(C.c_uint64)
    L.mgx_comm_last_error.restype = C.c_char_p
    sig = {
        "mgx_comm_unique_id": [vp], "mgx_comm_create": [vp, C.c_int, C.c_int, vp, pvp],
        "mgx_comm_create_all": [C.c_int, pvp, pvp], "mgx_comm_destroy": [vp],
        "mgx_slab_exchange": [vp, vp, C.c_int], "mgx_slab_exchange_all": [C.c_int, pvp, pvp, C.c_int],
        "mgx_slab_exchange_begin": [vp, vp, C.c_int], "mgx_slab_exchange_end": [vp],
        "mgx_slab_exchange_begin_all": [C.c_int, pvp, pvp, C.c_int], "mgx_slab_exchange_end_all": [C.c_int, pvp],
        "mgx_slab_periodic_ghosts": [vp, vp, C.c_int, C.c_int],
        "mgx_slab_periodic_ghosts_all": [C.c_int, pvp, pvp, C.c_int, C.c_int],
        "mgx_allreduce_result": [vp, vp, pdbl, pdbl, pu64],
        "mgx_allreduce_result_all": [C.c_int, pvp, pvp, pdbl, pdbl, pu64],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    _lib = L
    return L


def _ck(rc):
    if rc != 0:
        raise lib.MgxError(rc, load().mgx_comm_last_error().decode())


class Comm:
    """one rank of a communicator"""

    def __init__(self, handle, ctx, rank, world):
        self.h, self.ctx, self.rank, self.world = handle, ctx, rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _ck(load().mgx_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def create(cls, ctx, rank: int, world: int, uid: bytes) -> "Comm":
        h = C.c_void_p()
        _ck(load().mgx_comm_create(ctx.h, int(rank), int(world), C.c_char_p(uid), C.byref(h)))
        return cls(h, ctx, rank, world)

    @classmethod
    def create_all(cls, ctxs) -> list:
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        out = (C.c_void_p * n)()
        _ck(load().mgx_comm_create_all(n, arr, out))
        return [cls(C.c_void_p(out[i]), ctxs[i], i, n) for i in range(n)]

    def exchange(self, lv, planes: int = 2):
        _ck(load().mgx_slab_exchange(lv.h, self.h, int(planes)))

    def exchange_begin(self, lv, planes: int = 2):
        _ck(load().mgx_slab_exchange_begin(lv.h, self.h, int(planes)))

    def exchange_end(self):
        _ck(load().mgx_slab_exchange_end(self.h))

    def periodic_ghosts(self, lv, c0: bool, c1: bool):
        _ck(load().mgx_slab_periodic_ghosts(lv.h, self.h, int(c0), int(c1)))

    def allreduce_result(self, lv):
        e2, s = C.c_double(), C.c_double()
        limbs = (C.c_uint64 * 2)()
        _ck(load().mgx_allreduce_result(lv.h, self.h, C.byref(e2), C.byref(s), limbs))
        return e2.value, s.value, int(limbs[0]) | (int(limbs[1]) << 64)

    def close(self):
        if self.h:
            load().mgx_comm_destroy(self.h)
            self.h = None


def exchange_all(comms, lvs, planes: int = 2):
    n = len(comms)
    _ck(load().mgx_slab_exchange_all(n, (C.c_void_p * n)(*[l.h for l in lvs]), (C.c_void_p * n)(*[c.h for c in comms]),
                                     int(planes)))


def allreduce_result_all(comms, lvs):
    n = len(comms)
    e2, s = C.c_double(), C.c_double()
    limbs = (C.c_uint64 * 2)()
    _ck(load().mgx_allreduce_result_all(n, (C.c_void_p * n)(*[l.h for l in lvs]), (C.c_void_p * n)(*[c.h for c in comms]),
                                        C.byref(e2), C.byref(s), limbs))
    return e2.value, s.value, int(limbs[0]) | (int(limbs[1]) << 64)
