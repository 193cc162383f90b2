"""ad-hoc: is the first two-sweep pass after a stretch of small kernels slower than passes that follow one another?
Run under `rocprofv3 --kernel-trace`; the k_tb_sweep launches come in the order warm(6) A(5) B(3) C(3) D(3) E(3)."""
import time

import numpy as np

from multigrid_amd import lib

ctx = lib.Context(0)
ctx.set_kernel(3)
n = 512
def level(layer, n):
    lv = ctx.level(layer, (n, n, n), np.float64)
    lv.set_params(1.7, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
    q = n // 4
    lv.set_boxes([(q, 3 * q, q, 3 * q, 1, max(1, n // 20)), (q, 3 * q, q, 3 * q, n - n // 20, n)], [0.5, -0.5], [1.0, 1.0], [1, 1])
    lv.bring_back()
    lv.setup(True)
    return lv
top, child, small = level(1, n), level(2, n // 2), level(5, 32)
child.sweep(50, 0)
top.sweep(12, lib.RED_ERR)          # warm: 6 passes
ctx.sync()
top.sweep(10, lib.RED_ERR)          # A: 5 passes back to back
ctx.sync()
time.sleep(0.05)
top.sweep(6, lib.RED_ERR)           # B: after 50 ms of idling
ctx.sync()
for _ in range(300):
    small.sweep(1, lib.RED_ERR)
top.sweep(6, lib.RED_ERR)           # C: after 300 small launches
ctx.sync()
lib.prolong(child, top, 1)
top.sweep(6, lib.RED_ERR)           # D: after a prolongation into the level
ctx.sync()
for _ in range(300):
    small.sweep(1, lib.RED_ERR)
lib.prolong(child, top, 1)
top.sweep(6, lib.RED_ERR)           # E: small launches, then prolongation
ctx.sync()
