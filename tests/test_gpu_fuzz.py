"""GPU: a short randomised parity run (scripts/fuzz_parity.py: random shapes, boundary conditions, box tables,
magnitudes, kernels, sweep counts and reductions; random parent/child pairs for the transfers; levels cut into z
slabs at random planes; whole solves of random input files; the search for the cells of a box) against the oracle,
bit for bit.  The script runs thousands of cases on its own; the suite takes a fixed sample of them."""
import importlib.util
import os

import numpy as np
import pytest
import torch          # noqa: F401  before the HIP library: one HIP runtime per process

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("mode,cases,seed", [("sweeps", 150, 11), ("transfers", 150, 12), ("slabs", 80, 13), ("solves", 20, 14), ("boxes", 200, 15)])
def test_random_cases_equal_the_oracle(fuzz, mode, cases, seed):
    from multigrid_amd import lib
    rng = np.random.default_rng(seed)
    ctx = lib.Context(0)
    try:
        run = {"sweeps": fuzz.one_case, "transfers": fuzz.one_transfer, "slabs": fuzz.one_slab_case,
               "solves": fuzz.one_solve, "boxes": fuzz.one_box_place}[mode]
        for i in range(cases):
            run(ctx, rng, i)
    finally:
        ctx.close()
