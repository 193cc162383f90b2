"""The C-ABI shared library must load without a GPU and export every function include/mgx.h declares; the product
path has no CPU fallback (a missing library is an error, not a detour through the oracle)."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mgx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_entry_points():
    names = declared_functions()
    for must in ("mgx_level_sweep", "mgx_restrict", "mgx_prolong", "mgx_level_setup", "mgx_level_abs_sum",
                 "mgx_level_create_slab", "mgx_level_plane_ptr"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from multigrid_amd import lib
    assert os.path.exists(lib.LIB_PATH), "build it first: make -C multigrid_amd/csrc"
    dll = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(dll, n)]
    assert not missing, missing
    assert sorted(lib.SYMBOLS) == declared_functions()       # the ctypes binding covers the whole header
    dll.mgx_version.restype = ctypes.c_char_p
    assert dll.mgx_version().decode().startswith("mgx")


def test_missing_library_is_an_error(monkeypatch):
    from multigrid_amd import lib
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libmgx.so")
    monkeypatch.setattr(lib, "_lib", None, raising=False)
    with pytest.raises((OSError, ImportError)):
        lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multigrid_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".f90", ".F90")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "mg_oracle" not in src, f
