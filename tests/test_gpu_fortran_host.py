"""GPU: the Fortran host (ISO_C_BINDING module + the reference's cycle control in Fortran, multigrid_amd/fortran)
drives libmgx.so and must reproduce, bit for bit, what the oracle's red-black solve of the same hierarchy gives:
residual history, sweep counts and the final potential."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multigrid_amd import mgio                                   # noqa: E402
from oracle import oracle as orc                                 # noqa: E402
from tests.helpers import DT, bits_equal, load, oracle_levels    # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [("solve_64_bu", "f64", None), ("solve_64x48x80_v_half", "f64", None), ("solve_64x48x80_v_half", "f32", None),
         ("solve_80_v", "f64", ("sor 1.8", "sor 1.6"))]


@pytest.mark.parametrize("name,prec,sub", CASES)
def test_fortran_host_matches_oracle(name, prec, sub, tmp_path):
    exe = os.path.join(ROOT, "multigrid_amd", "fortran", f"mg_gpu_{prec}")
    assert os.path.exists(exe), "build it first: make -C multigrid_amd/fortran"
    text = str(load(f"{name}_{prec}")["input"])
    if sub:
        assert sub[0] in text
        text = text.replace(*sub)
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=DT[prec])
    method = mgio.read_method(IO)
    V, hist, per = orc.solve(oracle_levels(top), method, DT[prec], order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
    case, out = str(tmp_path / "case.txt"), str(tmp_path / "out.bin")
    mgio.write_host_case(top, method, case)
    r = subprocess.run([exe, case, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert r.returncode == 0, r.stdout
    levels = list(top.levels())
    shape = tuple(int(v) + 2 for v in reversed(levels[0].n))
    h, per_f, W = mgio.read_host_result(out, len(levels), shape, DT[prec])
    assert per_f == per
    assert len(h["itt"]) == len(hist)
    for f in ("layer", "kind", "itt"):
        assert np.array_equal(h[f], hist[f]), f
    for f in ("tol", "sum", "err"):
        assert np.array_equal(h[f], hist[f]), f
    assert bits_equal(W, V)


def test_fortran_host_without_print_only_sums(tmp_path):
    """`noprint`: no print-only sums, inner legs of a cycle only enqueued; same convergence tests, counts and potential"""
    name, prec = "solve_80_v", "f64"
    exe = os.path.join(ROOT, "multigrid_amd", "fortran", f"mg_gpu_{prec}")
    assert os.path.exists(exe), "build it first: make -C multigrid_amd/fortran"
    text = str(load(f"{name}_{prec}")["input"]).replace("sor 1.8", "sor 1.6")
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=DT[prec])
    method = mgio.read_method(IO)
    V, hist, per = orc.solve(oracle_levels(top), method, DT[prec], order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
    keep = hist[hist["kind"] != 1]
    case, out = str(tmp_path / "case.txt"), str(tmp_path / "out.bin")
    mgio.write_host_case(top, method, case)
    r = subprocess.run([exe, case, out, "noprint"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "solve seconds" in r.stdout
    levels = list(top.levels())
    shape = tuple(int(v) + 2 for v in reversed(levels[0].n))
    h, per_f, W = mgio.read_host_result(out, len(levels), shape, DT[prec])
    assert per_f == per
    assert len(h["itt"]) == len(keep)
    for f in ("layer", "kind", "itt", "tol", "sum", "err"):
        assert np.array_equal(h[f], keep[f]), f
    assert bits_equal(W, V)
