"""GPU: grid_solve's loop in one launch (k_small_solve, levels of up to 64^3 cells) against the same loop as a launch
per sweep (MGX_SOLVE_LOOP=0) and against the host loop over single sweeps: history (tol, sum, max dV^2), sweep count
and field, bit for bit; levels at and beyond the size limit, odd extents, boxes with rho != 1, every boundary kind.
And the layout of the result record the library hands out on the device (mgx_level_result_device)."""
import os

import numpy as np
import pytest
import torch          # noqa: F401  before the HIP library: one HIP runtime per process

from multigrid_amd import lib

pytestmark = pytest.mark.gpu

CASES = [
    # shape, bc, boxes (place, value, rho, constant)
    ((64, 64, 64), (2, 2, 2, 2, 2, 2), [((10, 20, 10, 50, 8, 9), 0.5, 1.0, 1), ((40, 50, 10, 50, 50, 51), -0.5, 1.0, 1)]),
    ((33, 17, 9), (1, 1, 3, 2, 2, 3), [((5, 9, 4, 12, 2, 3), 1.0, 1.0, 1), ((15, 25, 3, 14, 4, 8), 0.0, 3.0, 0)]),
    ((60, 37, 60), (2, 2, 2, 2, 3, 2), [((10, 50, 5, 30, 1, 2), 0.5, 1.0, 1), ((20, 40, 10, 25, 20, 40), 0.0, 2.5, 0),
                                          ((30, 35, 12, 20, 25, 30), -0.25, 1.0, 1)]),
    ((7, 5, 3), (2, 3, 2, 3, 1, 1), []),
    ((66, 64, 64), (2, 2, 1, 1, 2, 2), [((10, 20, 10, 50, 8, 9), 0.5, 1.0, 1)]),      # beyond the limit: launches per sweep
]


def _level(ctx, shape, dtype, bc, boxes, V0):
    lv = ctx.level(1, shape, dtype)
    lv.set_params(1.6, (0.3, 0.3, 0.4), bc)
    lv.set_boxes([b[0] for b in boxes], [b[1] for b in boxes], [b[2] for b in boxes], [b[3] for b in boxes])
    lv.upload(V0)
    lv.setup(False)
    return lv


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_one_launch_loop_equals_launch_per_sweep_and_host_loop(case, dtype):
    shape, bc, boxes = CASES[case]
    rng = np.random.default_rng(case)
    t = np.dtype(dtype).type
    runs = {}
    for mode in ("1", "0", "host"):
        os.environ["MGX_SOLVE_LOOP"] = "0" if mode == "0" else "1"
        ctx = lib.Context(0)                 # (the switch is read when a context is made)
        try:
            probe = ctx.level(1, shape, dtype)
            V0 = rng.standard_normal(probe.shape).astype(dtype) if "V0" not in runs else runs["V0"]
            runs["V0"] = V0
            probe.close()
            lv = _level(ctx, shape, dtype, bc, boxes, V0)
            count = lv.count_nonconst()
            old = t(lv.abs_sum())
            tol_level = 2.0e-4
            if mode == "host":               # the reference's loop (m_gs_CDS.F90:200-216) over single sweeps
                nr = t(1) / t(count)
                hist = []
                while True:
                    lv.sweep(1, lib.RED_ERR | lib.RED_SUM)
                    e2, s, _ = lv.result()
                    new = t(s)
                    tol = t(abs(t(old - new)) * nr)
                    hist.append((float(tol), float(new), e2))
                    old = new
                    if not (float(tol) > tol_level) or len(hist) >= 400:
                        break
                runs[mode] = (np.array(hist), len(hist), lv.download(), lv.result())
            else:
                h, n, st = lv.solve(tol_level, count, float(old), 400)
                assert st in (0, 1)
                runs[mode] = (h, n, lv.download(), lv.result())
            lv.close()
        finally:
            ctx.close()
            os.environ.pop("MGX_SOLVE_LOOP", None)
    ref = runs["host"]
    assert 1 < ref[1] <= 400
    for mode in ("1", "0"):
        h, n, V, res = runs[mode]
        assert n == ref[1], (mode, n, ref[1])
        assert np.array_equal(h.view(np.uint64), ref[0].view(np.uint64)), mode
        assert np.array_equal(V.view(np.uint8), ref[2].view(np.uint8)), mode
        assert res == ref[3], mode


def test_result_record_on_the_device_matches_the_host_record():
    from multigrid_amd import slab
    ctx = lib.Context(0)
    try:
        shape = (96, 50, 40)
        lv = ctx.level(1, shape, np.float64)
        lv.set_params(1.5, (0.3, 0.3, 0.4), (2, 2, 2, 2, 2, 2))
        lv.set_boxes([(10, 30, 10, 30, 5, 9)], [0.75], [1.0], [1])
        lv.bring_back()
        lv.setup(True)
        for kernel in (lib.KERNEL_AUTO, lib.KERNEL_TWO_PASS):      # a kernel that folds its own record, one that does not
            ctx.set_kernel(kernel)
            lv.sweep(3, lib.RED_ERR | lib.RED_SUM)
            ptr = lv.result_device()
            ctx.sync()
            rec = torch.as_tensor(slab._DevicePlanes(ptr, 64), device="cuda:0").view(torch.int64).cpu().numpy()
            e2, s, units = lv.result()
            assert np.array([rec[0]], dtype=np.int64).view(np.float64)[0] == e2
            assert sum(int(rec[1 + k]) << (32 * k) for k in range(4)) == units and units > 0
            assert rec[5] == 0 and rec[6] == 0
        lv.close()
    finally:
        ctx.close()
