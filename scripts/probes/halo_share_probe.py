"""probe: HBM fetch of one fused sweep on a level of 512 x 512 x NZ cells (one z chunk when NZ <= 64: every block starts at
the same time), to see whether y-neighbouring tiles share their halo rows in L2 when they march in step.
usage: halo_share_probe.py NZ [sweeps]"""
import sys

import numpy as np

from multigrid_amd import lib

nz = int(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = lib.Context(0)
ctx.set_kernel(2)
lv = ctx.level(1, (512, 512, nz), np.float64)
lv.set_params(1.7, (1 / 3,) * 3, (1, 1, 1, 1, 3, 2))
lv.set_boxes([], [], [], [])
rng = np.random.default_rng(1)
lv.upload(rng.standard_normal(lv.shape).astype(np.float64))
lv.setup(False)
for _ in range(k):
    lv.sweep(1, 0)
ctx.sync()
print("cells", 512 * 512 * nz, "compulsory read bytes", 512 * 512 * nz * 8)
