"""ad-hoc: one middle z slab (512 x 512 x 512 of a 512 x 512 x 1536 level) on one GPU, no communication: time of a
whole sweep in one launch against the edge + inner launches the halo exchange is hidden behind"""
import sys
import time

import numpy as np

from multigrid_amd import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = lib.Context(0)
lv = ctx.level(1, (n, n, 3 * n), np.float64, n, n)
lv.set_params(1.7, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
q = n // 4
B = [(q, 3 * q, q, 3 * q, 1, max(1, 3 * n // 20)), (q, 3 * q, q, 3 * q, 3 * n - 3 * n // 20, 3 * n),
     (2 * n // 5, 3 * n // 5, 2 * n // 5, 3 * n // 5, 3 * n // 20 + 1, 3 * n - 3 * n // 20 - 1)]
lv.set_boxes(B, [0.5, -0.5, 0.0], [1.0, 1.0, 3.0], [1, 1, 0])
rng = np.random.default_rng(1)
lv.upload(rng.standard_normal(lv.shape).astype(np.float64))
lv.setup(False)
lv.halo_written()
red = lib.RED_ERR | lib.RED_SUM
for mode in ("one launch", "edge + inner"):
    for reps in (5, 30):
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            if mode == "one launch":
                lv.sweep(1, red)
            else:
                lv.sweep_part(lib.PART_EDGE, red)
                lv.sweep_part(lib.PART_INNER, red)
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{mode}: {ms:.3f} ms/sweep  {n ** 3 / ms / 1e6:.1f} GLUP/s", flush=True)
