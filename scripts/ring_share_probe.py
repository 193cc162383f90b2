"""ad-hoc (GPU, under rocprofv3 --pmc FETCH_SIZE): pair passes on a box-free level of the given shape, so that the bytes
fetched per pass can be set against the field size: how much of the ring of a tile is served by L2
   python scripts/ring_share_probe.py n1 n2 n3 [passes]"""
import sys

import numpy as np

from multigrid_amd import lib

n = tuple(int(v) for v in sys.argv[1:4])
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 6
ctx = lib.Context(0)
ctx.set_kernel(lib.KERNEL_PAIR)
lv = ctx.level(1, n, np.float64)
lv.set_params(1.7, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
lv.set_boxes([], [], [], [])
rng = np.random.default_rng(1)
lv.upload(rng.standard_normal(lv.shape))
lv.setup(False)
for _ in range(passes):
    lv.sweep(2, 0)
ctx.sync()
print("field bytes", n[0] * n[1] * n[2] * 8)
