"""The entry point that allocates nothing: mgx_level_abs_sum_noalloc (static device scratch, result to device memory)
against mgx_level_abs_sum_limbs, through the C ABI, on two calls in a row, the second on a smaller level, then on a
level beyond the stated limit.  Bar: the same 128-bit integer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_no_alloc_abs_sum_matches_and_rejects_oversize():
    from multigrid_amd import lib
    ctx = lib.Context(0)
    try:
        # three words of device memory for the result: the start of plane 1 of a small level (x = 0, 1, 2 of row 0)
        holder = ctx.level(9, (8, 8, 8), np.float64)
        out_ptr, _ = holder.plane_ptr(1, 1)

        def read_out():
            ctx.sync()
            return [int(v) for v in holder.download()[1, 0, 0:3].view(np.uint64)]
        rng = np.random.default_rng(5)
        for n, dt in (((150, 70, 90), np.float64), ((37, 21, 13), np.float32), ((64, 64, 64), np.float64)):
            lv = ctx.level(1, n, dt)
            V = (rng.standard_normal((n[2] + 2, n[1] + 2, n[0] + 2)) * np.exp(rng.uniform(-30, 3, (n[2] + 2, n[1] + 2, n[0] + 2)))).astype(dt)
            lv.upload(V)
            want = lv.abs_sum_limbs()
            lv.abs_sum_noalloc(out_ptr)
            got = read_out()
            assert got[2] == 0
            assert (got[1] << 64) | got[0] == want, (n, dt)
            lv.close()
        # a value beyond the exact range is counted, as the allocating entry point reports it
        lv = ctx.level(1, (8, 8, 8), np.float64)
        V = np.zeros((10, 10, 10)); V[3, 4, 5] = 2.0 ** 21; V[1, 1, 1] = np.inf
        lv.upload(V)
        lv.abs_sum_noalloc(out_ptr)
        assert read_out()[2] == 2
        with pytest.raises(lib.MgxError):
            lv.abs_sum_limbs()
        lv.close()
        # more rows than the scratch array holds partial sums for: refused, nothing launched
        big = ctx.level(1, (2, 1024, 513), np.float32)
        with pytest.raises(lib.MgxError) as e:
            big.abs_sum_noalloc(out_ptr)
        assert e.value.code == -5 and "rows" in str(e.value)
        big.close()
    finally:
        ctx.close()
