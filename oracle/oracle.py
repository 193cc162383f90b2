"""TEST INFRASTRUCTURE — ctypes/numpy front end of the CPU oracle (oracle/mg_oracle.c).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the shipped package ``multigrid_amd``
never does.  Fields are numpy arrays of shape (n3+2, n2+2, n1+2) in C order,
i.e. the reference's ghosted Fortran array V(0:n1+1,0:n2+1,0:n3+1) byte for
byte (x contiguous).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmg_oracle.so")

ORDER_LEX, ORDER_RB, ORDER_RB_RED_FIRST = 0, 1, 2
SUM_SEQ, SUM_EXACT = 0, 1
METHOD_BOTTOM_UP, METHOD_W = 1, 2
BC_PERIODIC, BC_DIRICHLET, BC_NEUMANN = 1, 2, 3


class _Box(C.Structure):
    _fields_ = [("place", C.c_int32 * 6), ("constant", C.c_int32), ("pad_", C.c_int32),
                ("val", C.c_double), ("rho", C.c_double)]


class _Level(C.Structure):
    _fields_ = [("n", C.c_int32 * 3), ("bc", C.c_int32 * 6), ("nbox", C.c_int32), ("steps", C.c_int32),
                ("res_method", C.c_int32), ("pro_method", C.c_int32), ("sor", C.c_double),
                ("a", C.c_double * 3), ("tol", C.c_double), ("box", C.POINTER(_Box))]


class _Hist(C.Structure):
    _fields_ = [("layer", C.c_int32), ("kind", C.c_int32), ("itt", C.c_int32), ("pad_", C.c_int32),
                ("tol", C.c_double), ("sum", C.c_double), ("err", C.c_double)]


HIST_DTYPE = np.dtype([("layer", "i4"), ("kind", "i4"), ("itt", "i4"), ("pad_", "i4"),
                       ("tol", "f8"), ("sum", "f8"), ("err", "f8")])


@dataclass
class Box:
    place: Tuple[int, int, int, int, int, int]  # xmin,xmax,ymin,ymax,zmin,zmax (1-based, inclusive)
    val: float = 0.0
    rho: float = 1.0
    constant: bool = False


@dataclass
class Level:
    n: Tuple[int, int, int]
    bc: Tuple[int, int, int, int, int, int] = (2, 2, 2, 2, 2, 2)
    sor: float = 1.8
    a: Tuple[float, float, float] = (1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0)
    tol: float = 1e-4
    steps: int = 2
    res_method: int = 1
    pro_method: int = 1
    boxes: List[Box] = field(default_factory=list)

    def shape(self):
        return (self.n[2] + 2, self.n[1] + 2, self.n[0] + 2)


def build(force: bool = False) -> str:
    """Compile oracle/mg_oracle.c (gcc) if the shared object is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("mg_oracle.c", "mg_oracle_impl.inc", "mg_oracle.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        for s, ct in (("f32", C.c_float), ("f64", C.c_double)):
            p = C.POINTER(ct)
            getattr(_lib, f"orc_setup_{s}").argtypes = [C.POINTER(_Level), p, C.c_int]
            getattr(_lib, f"orc_setup_{s}").restype = None
            for name in ("orc_sweep", "orc_half_sweep"):
                f = getattr(_lib, f"{name}_{s}")
                f.argtypes = [C.POINTER(_Level), p, C.c_int]
                f.restype = C.c_double
            f = getattr(_lib, f"orc_abs_sum_seq_{s}")
            f.argtypes = [C.POINTER(_Level), p]
            f.restype = C.c_double
            f = getattr(_lib, f"orc_abs_sum_exact_{s}")
            f.argtypes = [C.POINTER(_Level), p, C.POINTER(C.c_uint64)]
            f.restype = C.c_double
            f = getattr(_lib, f"orc_tolerance_{s}")
            f.argtypes = [C.POINTER(_Level)]
            f.restype = C.c_double
            for name in ("orc_restrict", "orc_prolong"):
                f = getattr(_lib, f"{name}_{s}")
                f.argtypes = [C.POINTER(_Level), p, C.POINTER(_Level), p]
                f.restype = None
            f = getattr(_lib, f"orc_solve_{s}")
            f.argtypes = [C.POINTER(_Level), C.c_int, C.c_int, C.c_int, C.c_int, p, C.POINTER(_Hist),
                          C.c_int64, C.POINTER(C.c_int64), C.c_int64]
            f.restype = C.c_int64
        pd_, pi_ = C.POINTER(C.c_double), C.POINTER(C.c_int)
        _lib.orc_box_place.argtypes = [pi_, pd_, pd_, pd_, pd_, pi_]
        _lib.orc_box_place.restype = None
        _lib.orc_count_nonconst.argtypes = [C.POINTER(_Level)]
        _lib.orc_count_nonconst.restype = C.c_int64
    return _lib


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"unsupported grid precision {dtype}")


def _cast(dtype, v: float) -> float:
    """the value v after assignment to a real(grid_p) variable"""
    return float(np.dtype(dtype).type(v))


def _mk(levels: Sequence[Level], dtype):
    """ctypes array of levels (+ keep-alive list of box arrays)"""
    arr = (_Level * len(levels))()
    keep = []
    for i, L in enumerate(levels):
        c = arr[i]
        c.n[:] = list(L.n)
        c.bc[:] = list(L.bc)
        c.nbox = len(L.boxes)
        c.steps = L.steps
        c.res_method = L.res_method
        c.pro_method = L.pro_method
        c.sor = L.sor
        c.a[:] = [_cast(dtype, x) for x in L.a]
        c.tol = L.tol
        boxes = (_Box * max(1, len(L.boxes)))()
        for j, b in enumerate(L.boxes):
            boxes[j].place[:] = list(b.place)
            boxes[j].constant = 1 if b.constant else 0
            boxes[j].val = _cast(dtype, b.val)
            boxes[j].rho = _cast(dtype, b.rho)
        keep.append(boxes)
        c.box = C.cast(boxes, C.POINTER(_Box))
    return arr, keep


def _ptr(V: np.ndarray):
    assert V.flags["C_CONTIGUOUS"]
    return V.ctypes.data_as(C.POINTER(C.c_float if V.dtype == np.float32 else C.c_double))


def _check(L: Level, V: np.ndarray):
    assert V.shape == L.shape(), (V.shape, L.shape())


def setup(L: Level, V: np.ndarray, init: bool) -> None:
    _check(L, V)
    arr, keep = _mk([L], V.dtype)
    getattr(lib(), f"orc_setup_{_sfx(V.dtype)}")(arr, _ptr(V), int(init))


def sweep(L: Level, V: np.ndarray, order: int = ORDER_RB) -> float:
    """one full sweep in place; returns max (vv-V)^2 (not yet square-rooted)"""
    _check(L, V)
    arr, keep = _mk([L], V.dtype)
    return getattr(lib(), f"orc_sweep_{_sfx(V.dtype)}")(arr, _ptr(V), order)


def half_sweep(L: Level, V: np.ndarray, parity: int) -> float:
    _check(L, V)
    arr, keep = _mk([L], V.dtype)
    return getattr(lib(), f"orc_half_sweep_{_sfx(V.dtype)}")(arr, _ptr(V), parity)


def abs_sum_seq(L: Level, V: np.ndarray) -> float:
    _check(L, V)
    arr, keep = _mk([L], V.dtype)
    return getattr(lib(), f"orc_abs_sum_seq_{_sfx(V.dtype)}")(arr, _ptr(V))


def abs_sum_exact(L: Level, V: np.ndarray) -> Tuple[float, int]:
    """(rounded double, exact 128-bit integer in units of 2^-72)"""
    _check(L, V)
    arr, keep = _mk([L], V.dtype)
    limbs = (C.c_uint64 * 2)()
    s = getattr(lib(), f"orc_abs_sum_exact_{_sfx(V.dtype)}")(arr, _ptr(V), limbs)
    return s, (int(limbs[1]) << 64) | int(limbs[0])


def box_place(n, dL, offset, llc, box_cell) -> np.ndarray:
    """grid_add_box's loop over the cells (t_mg.f90:359-395): xmin,xmax,ymin,ymax,zmin,zmax (1-based).
    dL and box_cell are 3 x 3 with the lattice vectors as columns."""
    nn = (C.c_int * 3)(*[int(v) for v in n])
    d = np.asarray(dL, dtype=np.float64).reshape(3, 3).ravel(order="F").copy()
    b = np.asarray(box_cell, dtype=np.float64).reshape(3, 3).ravel(order="F").copy()
    o = np.ascontiguousarray(offset, dtype=np.float64)
    l = np.ascontiguousarray(llc, dtype=np.float64)
    out = (C.c_int * 6)()
    pd = C.POINTER(C.c_double)
    lib().orc_box_place(nn, d.ctypes.data_as(pd), o.ctypes.data_as(pd), l.ctypes.data_as(pd), b.ctypes.data_as(pd), out)
    return np.array(list(out), dtype=np.int64)


def count_nonconst(L: Level) -> int:
    arr, keep = _mk([L], np.float64)
    return lib().orc_count_nonconst(arr)


def tolerance(L: Level, dtype) -> float:
    arr, keep = _mk([L], dtype)
    return getattr(lib(), f"orc_tolerance_{_sfx(dtype)}")(arr)


def restrict(P: Level, Vp: np.ndarray, Cl: Level, Vc: np.ndarray) -> None:
    _check(P, Vp)
    _check(Cl, Vc)
    arr, keep = _mk([P, Cl], Vp.dtype)
    getattr(lib(), f"orc_restrict_{_sfx(Vp.dtype)}")(C.byref(arr[0]), _ptr(Vp), C.byref(arr[1]), _ptr(Vc))


def prolong(Cl: Level, Vc: np.ndarray, P: Level, Vp: np.ndarray) -> None:
    _check(P, Vp)
    _check(Cl, Vc)
    arr, keep = _mk([Cl, P], Vp.dtype)
    getattr(lib(), f"orc_prolong_{_sfx(Vp.dtype)}")(C.byref(arr[0]), _ptr(Vc), C.byref(arr[1]), _ptr(Vp))


class SolveStopped(RuntimeError):
    """orc_solve ran into max_sweeps; .hist holds the history up to there, .sweeps the sweeps per level"""


def solve(levels: Sequence[Level], method: int, dtype, order: int = ORDER_RB, sum_mode: int = SUM_EXACT,
          hist_cap: int = 1 << 20, max_sweeps: int = 1 << 24):
    """mg_gs_cds restated; returns (Vtop, history structured array, sweeps per level)"""
    dtype = np.dtype(dtype)
    arr, keep = _mk(levels, dtype)
    V = np.zeros(levels[0].shape(), dtype=dtype)
    hist = np.zeros(hist_cap, dtype=HIST_DTYPE)
    per = (C.c_int64 * len(levels))()
    n = getattr(lib(), f"orc_solve_{_sfx(dtype)}")(
        arr, len(levels), method, order, sum_mode, _ptr(V),
        hist.ctypes.data_as(C.POINTER(_Hist)), hist_cap, per, max_sweeps)
    if n == -2:
        e = SolveStopped(f"orc_solve stopped after max_sweeps = {max_sweeps} sweeps without meeting its tolerance")
        e.hist = hist[hist["layer"] > 0].copy()       # the records written before the stop
        e.sweeps = [int(x) for x in per]
        raise e
    if n < 0:
        raise RuntimeError(f"orc_solve failed with {n}")
    if n > hist_cap:
        raise RuntimeError(f"history overflow: {n} > {hist_cap}")
    return V, hist[:n].copy(), [int(x) for x in per]
