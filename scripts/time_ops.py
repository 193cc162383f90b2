"""ad-hoc: device time of the level operators other than the sweep (setup, restriction, prolongation, abs-sum)"""
import sys
import time

import numpy as np

from multigrid_amd import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = lib.Context(0)
bc = (1, 1, 1, 1, 3, 2)
P = ctx.level(1, (n,) * 3, np.float64)
Cc = ctx.level(2, (n // 2,) * 3, np.float64)
for lv in (P, Cc):
    lv.set_params(1.7, (1 / 3,) * 3, bc)
    lv.set_boxes(np.zeros((0, 6), np.int32), [], [], [])
    lv.bring_back()
    lv.setup(True)
rng = np.random.default_rng(0)
P.upload(rng.standard_normal(P.shape))
P.setup(False)


def t(label, f, reps, bytes_):
    f(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    ctx.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{label:28s} {ms:8.3f} ms   {bytes_ / ms / 1e6:8.1f} GB/s algorithmic", flush=True)


N = n ** 3
t("setup(init=False)", lambda: P.setup(False), 20, 0)
t("setup(init=True)", lambda: Cc.setup(True), 20, (n // 2 + 2) ** 3 * 8)
for m, name in ((1, "full"), (2, "half")):
    t(f"restrict {name}", lambda: lib.restrict(P, Cc, m), 20, 9 * N)
    t(f"prolong {name}", lambda: lib.prolong(Cc, P, m), 20, 9 * N)
t("abs_sum", lambda: P.abs_sum(), 20, 8 * N)
t("sweep reduce=3 (no boxes)", lambda: P.sweep(1, 3), 50, 16 * N)
ctx.set_kernel(1)
t("two-pass sweep reduce=3", lambda: P.sweep(1, 3), 20, 16 * N)
