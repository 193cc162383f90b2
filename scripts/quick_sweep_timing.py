"""ad-hoc: time red-black sweeps of one level on the GPU (not the official bench)"""
import os
import sys
import time

import numpy as np

from multigrid_amd import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dt = np.float64 if (len(sys.argv) <= 3 or sys.argv[3] == "f64") else np.float32
red = int(sys.argv[4]) if len(sys.argv) > 4 else 1
boxes = sys.argv[5] if len(sys.argv) > 5 else 'all'
ctx = lib.Context(0)
ctx.set_kernel(kernel)
lv = ctx.level(1, (n, n, n), dt)
lv.set_params(1.8, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
q = n // 4
B = [(q, 3 * q, q, 3 * q, 1, max(1, n // 20)), (q, 3 * q, q, 3 * q, n - n // 20, n),
     (2 * n // 5, 3 * n // 5, 2 * n // 5, 3 * n // 5, n // 20 + 1, n - n // 20 - 1)]
nbx = {'all': 3, 'const': 2, 'none': 0}[boxes]
lv.set_boxes(B[:nbx], [0.5, -0.5, 0.0][:nbx], [1.0, 1.0, 3.0][:nbx], [1, 1, 0][:nbx])
if nbx == 0:
    rng = np.random.default_rng(1)
    lv.upload(rng.standard_normal(lv.shape).astype(dt))
lv.setup(nbx > 0)
lv.sweep(3, red)
lv.result()
for reps in (10, 30):
    ctx.sync()
    t0 = time.perf_counter()
    if os.environ.get("PAIR"):          # two sweeps per call (the form of gs_V's legs), reductions on the second
        for _ in range(reps // 2):
            lv.sweep(2, red)
    elif os.environ.get("PER_SWEEP"):
        for _ in range(reps):
            lv.sweep(1, red)
    else:
        lv.sweep(reps, red)
    ctx.sync()
    t1 = time.perf_counter()
    ms = (t1 - t0) / reps * 1e3
    print(f"n={n} kernel={kernel} {np.dtype(dt).name} reduce={red} boxes={boxes}: {ms:.3f} ms/sweep  {n**3 / ms / 1e6:.1f} GLUP/s  "
          f"{n**3 * 2 * np.dtype(dt).itemsize / ms / 1e9:.3f} TB/s algorithmic", flush=True)
print(lv.result())
