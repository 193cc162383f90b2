"""ad-hoc: run-time ablations of the two-sweep kernel (needs a library built with -DMGX_TB_ABLATE_BITS)"""
import sys, time
import numpy as np
from multigrid_amd import lib
n = 512
ctx = lib.Context(0)
ctx.set_kernel(3)
lv = ctx.level(1, (n, n, n), np.float64)
lv.set_params(1.8, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
lv.set_boxes([], [], [], [])
rng = np.random.default_rng(1)
lv.upload(rng.standard_normal(lv.shape))
lv.setup(False)
lv.sweep(4, 0)
for name, bits in (("all", 0), ("no stores", 64), ("no loads", 128), ("no LDS writes", 256), ("no stores, no loads", 192),
                   ("no stores, loads, LDS writes", 448), ("all", 0)):
    lv.sweep(4, bits)
    ctx.sync()
    t0 = time.perf_counter()
    lv.sweep(40, bits)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name:32s} {ms:.3f} ms per pass", flush=True)
