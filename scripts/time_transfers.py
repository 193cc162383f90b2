"""ad-hoc (GPU): restriction 512^3 -> 256^3 and prolongation 256^3 -> 512^3 (fp64, full interpolation), time per call"""
import time

import numpy as np

from multigrid_amd import lib

ctx = lib.Context(0)
P = ctx.level(1, (512, 512, 512), np.float64)
Cc = ctx.level(2, (256, 256, 256), np.float64)
for lv in (P, Cc):
    lv.set_params(1.7, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
    lv.set_boxes([], [], [], [])
    lv.bring_back()
    lv.setup(True)
rng = np.random.default_rng(1)
P.upload(rng.standard_normal(P.shape))
P.setup(False)
for name, fn in (("restrict", lambda: lib.restrict(P, Cc, 1)), ("prolong", lambda: lib.prolong(Cc, P, 1))):
    for _ in range(3):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    ctx.sync()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call (with the set-up behind it)")
