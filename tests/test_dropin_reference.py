"""The boundary dropped into the reference itself: oracle/dropin_patch.py replaces the bodies of the reference's level
operators (grid_setup, grid_sum, grid_restriction, grid_prolongation, gs, ...) by calls into libmgx.so and
oracle/Makefile links the reference's own parser, cycle control and `mg` program over them
(oracle/_ref/mg_dropin_{f32,f64}).

CPU: the recipe builds where the reference and amdflang are present.  GPU: the patched reference program and this
package's `python -m multigrid_amd` solve the same input file; the .VMG files they write must hold the same bits
(same library, same cycle control, two different hosts: the reference's Fortran and this package's Python)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from multigrid_amd import mgio
from tests.helpers import DT, bits_equal, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MG_REFERENCE", "/root/reference")
FC = "/opt/rocm/bin/amdflang"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference's sources are not on this machine")
@pytest.mark.skipif(not os.path.exists(FC), reason="no amdflang")
def test_patched_reference_builds(tmp_path):
    if not os.path.exists(os.path.join(ROOT, "multigrid_amd", "libmgx.so")):
        pytest.skip("libmgx.so not built")
    # the patcher finds every routine it replaces and leaves the rest of the text alone
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "dropin_patch.py"), os.path.join(REF, "src"),
                        str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cds = (tmp_path / "m_gs_CDS.F90").read_text()
    orig = open(os.path.join(REF, "src", "m_gs_CDS.F90")).read()
    assert "mgx_level_sweep" in cds and "mgx_level_sweep" not in orig
    for keep in ("subroutine gs_V", "subroutine grid_solve", "subroutine gs_bottom_up", "do while"):
        assert cds.count(keep) == orig.count(keep) > 0
    tmg = (tmp_path / "t_mg.f90").read_text()
    for call in ("mgx_level_setup", "mgx_level_abs_sum", "mgx_level_count_nonconst", "mgx_level_bring_back",
                 "mgx_level_release", "mgx_level_set_boxes"):
        assert call in tmg
    itp = (tmp_path / "t_mg_interp.f90").read_text()
    assert "mgx_restrict" in itp and "mgx_prolong" in itp
    # and the whole recipe compiles and links
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "dropin"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for p in ("f32", "f64"):
        assert os.access(os.path.join(ROOT, "oracle", "_ref", f"mg_dropin_{p}"), os.X_OK)


@pytest.mark.gpu
@pytest.mark.parametrize("name,prec", [("solve_64_bu", "f64"), ("solve_64_bu", "f32"),
                                       ("solve_64x48x80_v_half", "f64"), ("solve_64x48x80_v_half", "f32")])
def test_patched_reference_program_equals_python_host(name, prec, tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", f"mg_dropin_{prec}")
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref/mg_dropin_* missing: run `make -C oracle dropin` where the reference is present")
    text = str(load(f"{name}_{prec}")["input"])
    out = {}
    for host in ("reference", "python"):
        d = tmp_path / host
        d.mkdir()
        (d / "mg.input").write_text(text)
        if host == "reference":
            cmd = [exe, "mg.input"]
        else:
            cmd = [sys.executable, "-m", "multigrid_amd"] + (["--double"] if prec == "f64" else []) + ["mg.input"]
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (host, r.stdout[-1500:], r.stderr[-1500:])
        out[host] = mgio.read_vmg(str(d / "out.VMG"), DT[prec])
    (n0, o0, c0, V0), (n1, o1, c1, V1) = out["reference"], out["python"]
    assert np.array_equal(n0, n1) and np.array_equal(o0, o1) and np.array_equal(c0, c1)
    finite, vmax, same = bool(np.isfinite(V0).all()), float(np.abs(V0).max()), bool(bits_equal(V0, V1))
    assert finite and vmax > 0.0
    assert same


@pytest.mark.gpu
@pytest.mark.parametrize("name,prec", [("solve_80_v", "f64"), ("solve_64_bu", "f32"), ("test6", "f64")])
def test_patched_reference_program_equals_the_reference_program(name, prec, tmp_path):
    """The reference's own `mg` program, compiled from its sources (oracle/_ref/mg_ref_*, on the host CPU, one thread),
    and the same program over libmgx.so with the sweeps in the reference's order (MGX_KERNEL=4) write the same .VMG
    file for the same input file, byte for byte: parser, hierarchy, box placement, cycle control are the reference's
    own text in both, every level operator runs on the GPU in the second.  solve_80_v is a v-cycle at the parser's
    default sor 1.8; test6 is test/test6.input."""
    exe_gpu = os.path.join(ROOT, "oracle", "_ref", f"mg_dropin_{prec}")
    exe_cpu = os.path.join(ROOT, "oracle", "_ref", f"mg_ref_{prec}")
    for e in (exe_gpu, exe_cpu):
        if not os.path.exists(e):
            pytest.fail(f"{e} missing: run `make -C oracle ref dropin` where the reference is present")
    text = str(load(f"{name}_{prec}")["input"])
    data = {}
    for host, exe in (("cpu", exe_cpu), ("gpu", exe_gpu)):
        d = tmp_path / host
        d.mkdir()
        (d / "mg.input").write_text(text)
        env = dict(os.environ, OMP_NUM_THREADS="1", MGX_KERNEL="4")
        r = subprocess.run([exe, "mg.input"], cwd=d, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (host, r.stdout[-1500:], r.stderr[-1500:])
        data[host] = (d / "out.VMG").read_bytes()
    assert len(data["cpu"]) > 1000
    assert data["cpu"] == data["gpu"]
