"""multigrid_amd — MI355X (gfx950) level operators of the zerothi/multigrid V-cycle behind a C ABI.

The compute path is the HIP library ``libmgx.so`` (sources in ``csrc/``, interface in
``include/mgx.h``).  This package is the Python host side: a ctypes binding (``lib``), a mirror of
the reference's level hierarchy (``grid``), its input grammar (``mgio``) and its cycle control
(``cycle``).  There is no CPU fallback: importing ``lib`` without the built library raises.
"""
__all__ = ["lib", "grid", "cycle", "mgio"]
