#!/usr/bin/env python3
"""CPU characterisation (oracle only, no GPU): for which over-relaxation factors does the reference's cycle
(restriction and prolongation overwrite the levels, m_gs_CDS.F90:220-294) converge, by sweep order?
Orders: the reference's lexicographic gs (m_gs_CDS.F90:317-345), red-black black-first (the order of the
m_gs_br sketch, :49-106, which the GPU path implements) and red-black red-first.
Writes a markdown table to stdout; run from the repository root:  python scripts/omega_scan.py [workers]"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OMEGAS = ["1.60", "1.65", "1.70", "1.72", "1.74", "1.76", "1.78", "1.80", "1.85"]
ORDERS = [("lexicographic", 0), ("red-black, black first", 1), ("red-black, red first", 2)]


def config3_small(n, sor, zbc):
    L = 0.1 * n
    z = {"mixed": "  bc c- neumann\n  bc c+ dirichlet", "periodic": "  bc c periodic", "dirichlet": "  bc c dirichlet"}[zbc]
    return f"""
cell {L:.10g} {L:.10g} {L:.10g}
max-layers 8
boxes 4
method v-cycle
begin layers
  sor {sor}
  tol 1.e-5
  v-steps 2
  mesh-size {n} {n} {n}
  bc a periodic
  bc b periodic
{z}
  interp full
end layers
begin box
  llc {0.4 * L:.10g} {0.4 * L:.10g} {0.05 * L:.10g}
  cell {0.2 * L:.10g} {0.2 * L:.10g} {0.9 * L:.10g}
  value 0.
  rho 3.
  constant false
end box
begin box
  llc {0.25 * L:.10g} {0.25 * L:.10g} 0.
  cell {0.5 * L:.10g} {0.5 * L:.10g} {0.05 * L:.10g}
  value 0.5
end box
begin box
  llc {0.25 * L:.10g} {0.25 * L:.10g} {0.95 * L:.10g}
  cell {0.5 * L:.10g} {0.5 * L:.10g} {0.05 * L:.10g}
  value -0.5
end box
"""


def case_text(case, sor):
    from tests.helpers import load
    if case == "solve_80_v":
        t = str(load("solve_80_v_f64")["input"])
        assert "sor 1.8" in t
        return t.replace("sor 1.8", f"sor {sor}")
    n, zbc = case.split(":")
    return config3_small(int(n), sor, zbc)


def run(job):
    case, sor, oname, order = job
    from multigrid_amd import mgio
    from oracle import oracle as orc
    from tests.helpers import oracle_levels
    IO = mgio.MgIO(text=case_text(case, sor))
    top = mgio.iomg_read(IO, dtype=np.float64)
    try:
        V, hist, per = orc.solve(oracle_levels(top), mgio.read_method(IO), np.float64, order=order,
                                 sum_mode=orc.SUM_SEQ, max_sweeps=4000)
    except orc.SolveStopped as e:
        # maximum principle: |V| never exceeds the largest box value (0.5 here), so sum|V| <= 0.5 N for any state of a
        # converging iteration; beyond N cells the iteration has left that range for good
        sums = e.hist["sum"]
        ncell = float(np.prod([int(v) for v in top.n]))
        if len(sums) and (not np.isfinite(sums[-1]) or abs(sums[-1]) > ncell):
            return case, sor, oname, f"DIVERGES (sum|V| {sums[-1]:.2g} > N = {ncell:.2g} after {sum(e.sweeps)} sweeps)"
        return case, sor, oname, f"no divergence, tolerance not met in {sum(e.sweeps)} sweeps"
    vmax = float(np.nanmax(np.abs(V))) if np.isfinite(V).any() else float("inf")
    if not np.isfinite(V).all() or vmax > 1e3:
        k = int(np.argmax(~np.isfinite(hist["sum"]) | (np.abs(hist["sum"]) > 1e3 * max(1.0, abs(hist["sum"][0])))))
        return case, sor, oname, f"DIVERGES (max|V| {vmax:.1e}; sum leaves range at record {k} of {len(hist)})"
    return case, sor, oname, f"converges, sweeps per level {per}"


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cases = ["solve_80_v", "96:mixed", "96:periodic", "96:dirichlet"]
    jobs = [(c, w, on, o) for c in cases for w in OMEGAS for on, o in ORDERS]
    with mp.Pool(workers) as pool:
        res = pool.map(run, jobs, chunksize=1)
    for c in cases:
        print(f"\n### {c}\n\n| sor | " + " | ".join(on for on, _ in ORDERS) + " |\n|---|" + "---|" * len(ORDERS))
        for w in OMEGAS:
            row = [r[3] for r in res if r[0] == c and r[1] == w]
            print(f"| {w} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
