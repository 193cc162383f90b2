"""ad-hoc (GPU): the one-launch reference-order sweep (MGX_LEX_TILED=2) against the launch-per-tile-hyperplane form
(MGX_LEX_TILED=1) on noise, many sweeps, bit for bit; and its time per sweep."""
import os
import sys
import time

import numpy as np

from multigrid_amd import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dt = np.float64 if (len(sys.argv) <= 3 or sys.argv[3] == "f64") else np.float32
shape = (n, n - 3, n + 5) if len(sys.argv) > 4 else (n, n, n)
q = n // 4
B = [(q, 3 * q, q, 3 * q, 1, max(1, n // 20)), (q, 3 * q, q, 3 * q, n - n // 20, n),
     (2 * n // 5, 3 * n // 5, 2 * n // 5, 3 * n // 5, n // 20 + 1, n - n // 20 - 1)]
rng = np.random.default_rng(3)
V0 = None
out = {}
for mode in ("1", "2"):
    os.environ["MGX_LEX_TILED"] = mode
    ctx = lib.Context(0)
    ctx.set_kernel(lib.KERNEL_LEX)
    lv = ctx.level(1, shape, dt)
    lv.set_params(1.8, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
    lv.set_boxes(B, [0.5, -0.5, 0.0], [1.0, 1.0, 3.0], [1, 1, 0])
    if V0 is None:
        V0 = rng.standard_normal(lv.shape).astype(dt)
    lv.upload(V0)
    lv.setup(False)
    lv.sweep(2, 1)
    r0 = lv.result()
    ctx.sync()
    t0 = time.perf_counter()
    lv.sweep(sweeps, 1)
    ctx.sync()
    ms = (time.perf_counter() - t0) / sweeps * 1e3
    res = lv.result()
    W = lv.download()
    out[mode] = (W, res)
    cells = shape[0] * shape[1] * shape[2]
    print(f"mode {mode}: {shape} {np.dtype(dt).name}: {ms:.3f} ms per sweep, {cells / ms / 1e6:.1f} GLUP/s, result {res[0]:.6e}", flush=True)
    ctx.close()
iv = np.uint64 if dt == np.float64 else np.uint32
same = np.array_equal(out["1"][0].view(iv), out["2"][0].view(iv)) and out["1"][1] == out["2"][1]
print("identical:", same)
sys.exit(0 if same else 1)
