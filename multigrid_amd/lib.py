"""ctypes binding of the C ABI in include/mgx.h.

Fields cross the boundary as numpy arrays of shape (n3+2, n2+2, n1+2) in C order, which is the
reference's ghosted Fortran array V(0:n1+1,0:n2+1,0:n3+1) byte for byte (src/t_mg.f90:608).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGX_LIB") or os.path.join(_HERE, "libmgx.so")

RED_ERR, RED_SUM = 1, 2
KERNEL_AUTO, KERNEL_TWO_PASS, KERNEL_FUSED, KERNEL_PAIR = 0, 1, 2, 3
KERNEL_LEX = 4      # the reference's own sweep order (results of the reference on one thread, bit for bit; slow)
INTERP_FULL, INTERP_HALF = 1, 2
BC_PERIODIC, BC_DIRICHLET, BC_NEUMANN = 1, 2, 3
PART_EDGE, PART_INNER = 1, 2
PART_EDGE2, PART_INNER2 = 3, 4        # the two launches of a pass of two sweeps

SYMBOLS = [
    "mgx_last_error", "mgx_version", "mgx_ctx_create", "mgx_ctx_destroy", "mgx_ctx_sync", "mgx_ctx_set_kernel",
    "mgx_ctx_info", "mgx_level_context",
    "mgx_ctx_capture_begin", "mgx_ctx_capture_end", "mgx_graph_launch", "mgx_graph_destroy",
    "mgx_level_create", "mgx_level_create_slab", "mgx_level_destroy", "mgx_level_bring_back", "mgx_level_release",
    "mgx_level_set_params", "mgx_level_set_boxes", "mgx_level_setup", "mgx_level_sweep", "mgx_level_sweep_part",
    "mgx_level_half_sweep", "mgx_level_pair_pass",
    "mgx_level_result_reset", "mgx_level_result", "mgx_level_result_device", "mgx_level_abs_sum", "mgx_level_abs_sum_limbs",
    "mgx_level_abs_sum_noalloc", "mgx_level_solve",
    "mgx_limbs_to_double", "mgx_level_count_nonconst", "mgx_level_tolerance", "mgx_restrict", "mgx_prolong",
    "mgx_level_upload", "mgx_level_download", "mgx_level_plane_ptr", "mgx_level_halo_written", "mgx_level_info",
    "mgx_box_place",
]


class MgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mgx error {code}: {msg}")
        self.code = code


_lib = None
# PyTorch-ROCm ships a HIP runtime of its own.  A process that uses both must import torch before this library is
# loaded, so that the library binds to torch's runtime; the other order leaves two runtimes, and torch finds no GPU.
torch_loaded_first = False


def load():
    """Load libmgx.so.  Raises if the HIP library has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C multigrid_amd/csrc` "
                          "(or __graft_entry__.build()); multigrid_amd has no CPU fallback")
    global torch_loaded_first
    torch_loaded_first = "torch" in sys.modules
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pi32, pdbl, pu64 = C.POINTER(i32), C.POINTER(dbl), C.POINTER(C.c_uint64)
    L.mgx_last_error.restype = C.c_char_p
    L.mgx_version.restype = C.c_char_p
    L.mgx_limbs_to_double.restype = dbl
    L.mgx_limbs_to_double.argtypes = [pu64]
    sig = {
        "mgx_ctx_create": [C.c_int, vp, C.POINTER(vp)],
        "mgx_ctx_destroy": [vp], "mgx_ctx_sync": [vp], "mgx_ctx_set_kernel": [vp, C.c_int],
        "mgx_ctx_info": [vp, C.POINTER(C.c_int), C.POINTER(vp)], "mgx_level_context": [vp, C.POINTER(vp)],
        "mgx_ctx_capture_begin": [vp], "mgx_ctx_capture_end": [vp, C.POINTER(vp)],
        "mgx_graph_launch": [vp], "mgx_graph_destroy": [vp],
        "mgx_level_create": [vp, C.c_int, pi32, C.c_int, C.POINTER(vp)],
        "mgx_level_create_slab": [vp, C.c_int, pi32, i32, i32, C.c_int, C.POINTER(vp)],
        "mgx_level_destroy": [vp], "mgx_level_bring_back": [vp], "mgx_level_release": [vp],
        "mgx_level_set_params": [vp, dbl, pdbl, pi32],
        "mgx_level_set_boxes": [vp, i32, pi32, pdbl, pdbl, pi32],
        "mgx_level_setup": [vp, C.c_int],
        "mgx_level_sweep": [vp, C.c_int, C.c_int],
        "mgx_level_sweep_part": [vp, C.c_int, C.c_int],
        "mgx_level_half_sweep": [vp, C.c_int, i32, i32, C.c_int],
        "mgx_level_pair_pass": [vp, pi32],
        "mgx_level_result_reset": [vp],
        "mgx_level_result": [vp, pdbl, pdbl, pu64],
        "mgx_level_result_device": [vp, C.POINTER(vp)],
        "mgx_level_abs_sum": [vp, pdbl], "mgx_level_abs_sum_limbs": [vp, pu64],
        "mgx_level_abs_sum_noalloc": [vp, vp],
        "mgx_level_solve": [vp, dbl, i64, dbl, i64, i64, pdbl, C.POINTER(i64), pi32],
        "mgx_level_count_nonconst": [vp, C.POINTER(i64)],
        "mgx_level_tolerance": [vp, dbl, pdbl],
        "mgx_restrict": [vp, vp, C.c_int], "mgx_prolong": [vp, vp, C.c_int],
        "mgx_level_upload": [vp, vp], "mgx_level_download": [vp, vp],
        "mgx_level_plane_ptr": [vp, i32, i32, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "mgx_level_halo_written": [vp],
        "mgx_level_info": [vp, pi32, pi32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
        "mgx_box_place": [vp, pi32, pdbl, pdbl, pdbl, pdbl, pi32],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    _lib = L
    return L


_shared = {}


def default_device() -> int:
    """the device of host-side helpers that are not handed a context: MGX_DEVICE, else the rank's own device in a
    one-process-per-GPU job (LOCAL_RANK), else 0"""
    for name in ("MGX_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(name)
        if v is not None and v.strip().lstrip("-").isdigit():
            return max(0, int(v))
    return 0


def shared_context(device: Optional[int] = None) -> "Context":
    """a context (own stream) for host-side helpers that need the device before any solver exists (box placement on
    skewed cells while an input file is read): one per device, closed when the process ends"""
    device = default_device() if device is None else int(device)
    if device not in _shared:
        if not _shared:
            import atexit
            atexit.register(close_shared_contexts)
        _shared[device] = Context(device)
    return _shared[device]


def close_shared_contexts():
    for ctx in list(_shared.values()):
        try:
            ctx.close()
        except Exception:
            pass
    _shared.clear()


def _ck(rc):
    if rc != 0:
        raise MgxError(rc, load().mgx_last_error().decode())


class Graph:
    """a recorded sequence of library calls (mgx_graph)"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def launch(self):
        _ck(self.ctx.L.mgx_graph_launch(self.h))

    def close(self):
        if self.h:
            self.ctx.L.mgx_graph_destroy(self.h)
            self.h = None


class Context:
    """one GPU + one stream (mgx_ctx).  `stream` = raw hipStream_t value (e.g. torch's cuda_stream) or None."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.L = load()
        h = C.c_void_p()
        _ck(self.L.mgx_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device
        self.stream = int(stream) if stream else None   # None: the library enqueues on a stream of its own
        self._levels = []
        self._graphs = []

    def set_kernel(self, which: int):
        _ck(self.L.mgx_ctx_set_kernel(self.h, which))

    def sync(self):
        _ck(self.L.mgx_ctx_sync(self.h))

    def capture_begin(self):
        """record the calls that follow into a HIP graph instead of running them"""
        _ck(self.L.mgx_ctx_capture_begin(self.h))

    def capture_end(self, keep: bool = True) -> Optional["Graph"]:
        """end the recording; the Graph (not yet run: launch it), or None when dropped (keep=False)"""
        if not keep:
            _ck(self.L.mgx_ctx_capture_end(self.h, None))
            return None
        g = C.c_void_p()
        _ck(self.L.mgx_ctx_capture_end(self.h, C.byref(g)))
        gr = Graph(self, g)
        self._graphs.append(gr)
        return gr

    def box_place(self, n: Sequence[int], dL, offset, llc, box_cell) -> np.ndarray:
        """grid_add_box's loop over the cells of a level (t_mg.f90:359-395) on the device: xmin,xmax,ymin,ymax,zmin,zmax.
        dL and box_cell are 3 x 3 with the lattice vectors as columns (dL[:, 0] = dL(:,1))."""
        n3 = (C.c_int32 * 3)(*[int(v) for v in n])
        d = np.asfortranarray(np.asarray(dL, dtype=np.float64).reshape(3, 3)).ravel(order="F")
        b = np.asfortranarray(np.asarray(box_cell, dtype=np.float64).reshape(3, 3)).ravel(order="F")
        o = np.ascontiguousarray(offset, dtype=np.float64)
        l = np.ascontiguousarray(llc, dtype=np.float64)
        out = (C.c_int32 * 6)()
        pd = C.POINTER(C.c_double)
        _ck(self.L.mgx_box_place(self.h, n3, d.ctypes.data_as(pd), o.ctypes.data_as(pd), l.ctypes.data_as(pd),
                                 b.ctypes.data_as(pd), out))
        return np.array(list(out), dtype=np.int64)

    def level(self, layer: int, n: Sequence[int], dtype, z0: int = 0, n3_local: Optional[int] = None) -> "DeviceLevel":
        lv = DeviceLevel(self, layer, n, dtype, z0, n3_local)
        self._levels.append(lv)
        return lv

    def close(self):
        for gr in self._graphs:
            gr.close()
        self._graphs = []
        for lv in self._levels:
            lv.close()
        self._levels = []
        if self.h:
            self.L.mgx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceLevel:
    """a level resident in HBM (mgx_level)"""

    def __init__(self, ctx: Context, layer: int, n: Sequence[int], dtype, z0: int = 0, n3_local: Optional[int] = None):
        self.ctx, self.L = ctx, ctx.L
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError(f"grid precision {dtype} not supported")
        self.n_global = tuple(int(v) for v in n)
        self.z0 = int(z0)
        self.n3_local = int(n3_local) if n3_local is not None else self.n_global[2]
        self.n = (self.n_global[0], self.n_global[1], self.n3_local)
        self.layer = layer
        nn = (C.c_int32 * 3)(*self.n_global)
        h = C.c_void_p()
        _ck(self.L.mgx_level_create_slab(ctx.h, layer, nn, self.z0, self.n3_local, self.dtype.itemsize, C.byref(h)))
        self.h = h

    @property
    def shape(self) -> Tuple[int, int, int]:
        return (self.n[2] + 2, self.n[1] + 2, self.n[0] + 2)

    def close(self):
        if self.h:
            self.L.mgx_level_destroy(self.h)
            self.h = None

    def set_params(self, sor: float, a: Sequence[float], bc: Sequence[int]):
        _ck(self.L.mgx_level_set_params(self.h, float(sor), (C.c_double * 3)(*[float(x) for x in a]),
                                         (C.c_int32 * 6)(*[int(x) for x in bc])))

    def set_boxes(self, place, val, rho, constant):
        place = np.ascontiguousarray(place, dtype=np.int32).reshape(-1, 6)
        nb = place.shape[0]
        val = np.ascontiguousarray(val, dtype=np.float64).reshape(nb)
        rho = np.ascontiguousarray(rho, dtype=np.float64).reshape(nb)
        constant = np.ascontiguousarray(constant, dtype=np.int32).reshape(nb)
        _ck(self.L.mgx_level_set_boxes(
            self.h, nb, place.ctypes.data_as(C.POINTER(C.c_int32)), val.ctypes.data_as(C.POINTER(C.c_double)),
            rho.ctypes.data_as(C.POINTER(C.c_double)), constant.ctypes.data_as(C.POINTER(C.c_int32))))

    def bring_back(self):
        _ck(self.L.mgx_level_bring_back(self.h))

    def release(self):
        _ck(self.L.mgx_level_release(self.h))

    def setup(self, init: bool):
        _ck(self.L.mgx_level_setup(self.h, int(bool(init))))

    def sweep(self, nsweeps: int = 1, reduce: int = RED_ERR | RED_SUM):
        _ck(self.L.mgx_level_sweep(self.h, int(nsweeps), int(reduce)))

    def sweep_part(self, part: int, reduce: int = RED_ERR | RED_SUM):
        _ck(self.L.mgx_level_sweep_part(self.h, int(part), int(reduce)))

    def pair_pass(self) -> bool:
        """does sweep(2) take the two sweeps in one pass (undivided level) / do such passes pay (slab)?"""
        yes = C.c_int32()
        _ck(self.L.mgx_level_pair_pass(self.h, C.byref(yes)))
        return bool(yes.value)

    def half_sweep(self, parity: int, zlo: int, zhi: int, reduce: int = 0):
        _ck(self.L.mgx_level_half_sweep(self.h, int(parity), int(zlo), int(zhi), int(reduce)))

    def result_device(self) -> int:
        """device address of the eight-word record of the last sweep (mgx_level_result_device); nothing waits"""
        p = C.c_void_p()
        _ck(self.L.mgx_level_result_device(self.h, C.byref(p)))
        return int(p.value)

    def result_reset(self):
        _ck(self.L.mgx_level_result_reset(self.h))

    def result(self):
        """(max (vv-V)^2, sum|V| as double, exact sum as python int in units of 2^-72)"""
        e, s = C.c_double(), C.c_double()
        limbs = (C.c_uint64 * 2)()
        _ck(self.L.mgx_level_result(self.h, C.byref(e), C.byref(s), limbs))
        return e.value, s.value, (int(limbs[1]) << 64) | int(limbs[0])

    def abs_sum(self) -> float:
        s = C.c_double()
        _ck(self.L.mgx_level_abs_sum(self.h, C.byref(s)))
        return s.value

    def abs_sum_limbs(self) -> int:
        limbs = (C.c_uint64 * 2)()
        _ck(self.L.mgx_level_abs_sum_limbs(self.h, limbs))
        return (int(limbs[1]) << 64) | int(limbs[0])

    def solve(self, tol_level: float, count: int, old_sum: float, max_sweeps: int):
        """grid_solve's loop with the convergence test on the device: (history array [n][3] of tol, sum, max dV^2;
        status 0 converged / 1 max_sweeps / 2 sum out of range)"""
        cap = int(max(1, max_sweeps))       # every sweep gets its row (callers ask for chunks of sweeps)
        hist = np.zeros((cap, 3), dtype=np.float64)
        n, st = C.c_int64(), C.c_int32()
        _ck(self.L.mgx_level_solve(self.h, float(tol_level), int(count), float(old_sum), int(max_sweeps), cap,
                                   hist.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n), C.byref(st)))
        return hist[:min(n.value, cap)], int(n.value), int(st.value)

    def abs_sum_noalloc(self, out_device_ptr: int):
        """enqueue grid_sum without allocating or waiting; the three result words land at the DEVICE address given"""
        _ck(self.L.mgx_level_abs_sum_noalloc(self.h, C.c_void_p(int(out_device_ptr))))

    def count_nonconst(self) -> int:
        n = C.c_int64()
        _ck(self.L.mgx_level_count_nonconst(self.h, C.byref(n)))
        return n.value

    def tolerance(self, tol: float) -> float:
        out = C.c_double()
        _ck(self.L.mgx_level_tolerance(self.h, float(tol), C.byref(out)))
        return out.value

    def upload(self, V: np.ndarray):
        V = np.ascontiguousarray(V, dtype=self.dtype)
        if V.shape != self.shape:
            raise ValueError(f"field shape {V.shape}, level wants {self.shape}")
        _ck(self.L.mgx_level_upload(self.h, V.ctypes.data_as(C.c_void_p)))

    def download(self) -> np.ndarray:
        V = np.empty(self.shape, dtype=self.dtype)
        _ck(self.L.mgx_level_download(self.h, V.ctypes.data_as(C.c_void_p)))
        return V

    def plane_ptr(self, z: int, count: int = 1) -> Tuple[int, int]:
        p, b = C.c_void_p(), C.c_size_t()
        _ck(self.L.mgx_level_plane_ptr(self.h, int(z), int(count), C.byref(p), C.byref(b)))
        return p.value, b.value

    def halo_written(self):
        _ck(self.L.mgx_level_halo_written(self.h))

    def info(self):
        n = (C.c_int32 * 3)()
        eb = C.c_int32()
        px, pz, db = C.c_int64(), C.c_int64(), C.c_int64()
        _ck(self.L.mgx_level_info(self.h, n, C.byref(eb), C.byref(px), C.byref(pz), C.byref(db)))
        return dict(n=tuple(n), elem_bytes=eb.value, pitch_x=px.value, pitch_plane=pz.value, device_bytes=db.value)


def limbs_to_double(v: int) -> float:
    limbs = (C.c_uint64 * 2)(v & 0xFFFFFFFFFFFFFFFF, v >> 64)
    return load().mgx_limbs_to_double(limbs)


def restrict(parent: DeviceLevel, child: DeviceLevel, method: int = INTERP_FULL):
    _ck(parent.L.mgx_restrict(parent.h, child.h, int(method)))


def prolong(child: DeviceLevel, parent: DeviceLevel, method: int = INTERP_FULL):
    _ck(child.L.mgx_prolong(child.h, parent.h, int(method)))
