"""`python -m multigrid_amd [--double] [input]` — the reference's `mg` program (src/mg.F90:58-135) over the HIP library:
read the input file (default mg.input), take the `method` line, write the `init-save` files from the set-up state,
solve on the GPU, write the `save` files (default mg.vmg).  Files are the binary .VMG container
(src/m_mg_save.F90:231-267); the file type follows the reference's rule on the extension (:45-62: everything that
is not `cube` is binary; `vmgasc` is shadowed by `vmg`), and the cube writer is not part of this package.
`--double` runs in real(8) (the reference as shipped is real(4), t_mg.f90:8).  `--reference-order` sweeps in the
reference's own order (gs, m_gs_CDS.F90:317-345) instead of red-black: the results of the reference on one thread,
bit for bit in real(8), at any over-relaxation, several times slower.  Without a flag that order is selected by itself
for `method v-cycle` inputs whose over-relaxation the red-black ordering does not take (sor > 1.7; the parser's
default 1.8, m_mg_io.F90:87, among them), with one line on stderr; `--red-black` keeps the red-black kernels."""
from __future__ import annotations

import sys

import numpy as np

from . import cycle, mgio
from . import grid as G


def mg_save(top: G.MgGrid, filename: str, V: np.ndarray) -> str:
    i = filename.find(".")
    if i >= 0 and filename[i + 1:i + 5].lower() == "cube":
        raise G.MgStop("cube files are not written by this package: use a .vmg name")
    path = mgio.vmg_name(filename)
    mgio.write_vmg(path, top, V)
    return path


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    precision = np.float64 if "--double" in argv else np.float32
    order = (cycle.ORDER_REFERENCE if "--reference-order" in argv else
             cycle.ORDER_RED_BLACK if "--red-black" in argv else cycle.ORDER_AUTO)
    argv = [a for a in argv if not a.startswith("--")]
    filein = argv[-1] if argv else "mg.input"        # the last argument, as in mg.F90:60-66
    IO = mgio.MgIO(path=filein)
    top = mgio.iomg_read(IO, dtype=precision)
    method = mgio.read_method(IO)
    if order == cycle.ORDER_RED_BLACK and method == cycle.CDS_W_CYCLE and any(float(g.sor) > cycle.RB_SOR_LIMIT for g in top.levels()):
        print("warning: `method v-cycle` with sor > 1.7." + cycle.divergence_hint(max(float(g.sor) for g in top.levels())),
              file=sys.stderr)
    solver = cycle.Solver(top, verbose=True, order=order)
    try:
        names = mgio.save_names(IO, "init-save")
        if names:
            solver.bring_back(top)
            solver.setup(top, True)
            V = top.dev.download()
            for name in names:
                print("Saved", mg_save(top, name, V))
            solver.hold_back(top)
        solver.mg_gs_cds(method)
        V = top.dev.download()
        for name in mgio.save_names(IO, "save") or ["mg.vmg"]:
            print("Saved", mg_save(top, name, V))
    except G.MgStop as e:
        print(e, file=sys.stderr)
        return 1
    finally:
        solver.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
