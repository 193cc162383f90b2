#!/usr/bin/env python3
"""Benchmark of the V-cycle hot path on MI355X.

--gpus 1 (default): one "step" = one full red-black SOR sweep (black + red half-sweeps, with the fused max-dV^2
and sum|V| reductions that the host convergence loop reads, m_gs_CDS.F90:200-210) of the finest level of the
512^3 mixed-boundary configuration (config 3), fp64, field resident in HBM.  `value` = lattice updates per second.
--gpus N > 1: the 2048^3 periodic configuration (config 5; --config 4 selects 256 x 256 x 4096) decomposed into N
z slabs, strong scaling: every rank sweeps its slab, two halo planes per neighbour travel per sweep over RCCL
beside the bulk of the sweep, the reductions are all-reduced.  `value` = lattice updates per second of the whole job.

The JSON line also carries
  roofline      achieved HBM GB/s of the sweep kernel (16 B per lattice update: one read + one write of every cell per
                sweep) from HIP events on the kernel's stream, against 8 TB/s; sub-objects `on_solution` (the same
                launch on the converged potential) and `pair_pass` (two sweeps in one pass over the field, the form
                gs_V's legs run in: 16 B per cell per PASS)
  cpu_baseline  the reference's own smoother (compiled from the reference sources into oracle/_ref/, OpenMP) — or the
                C oracle if that binary is not present — timed on a bounded sample on this host, all cores and one
  vcycle        wall time of the complete `method v-cycle` solve of config 3 to its tolerance (tol 1e-8 unless
                --vcycle-tol), with sweep counts per level; with --gpus N > 1 the same solve with every level
                decomposed into z slabs over the N GPUs
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "V-cycle time-to-1e-8 residual at 512³; red-black sweep GLUP/s; % HBM peak"
HBM_PEAK_GBS = 8000.0
BYTES_PER_LUP = 16.0          # fp64: every cell read once and written once per sweep


def config3_input(n=(512, 512, 512), tol="1.e-8", sor="1.7", steps=2, method="v-cycle", max_layers=8) -> str:
    """512^3 mixed Dirichlet/Neumann/periodic case in the reference's input grammar
    (two plate electrodes +-0.5 and a rho = 3 block between them)"""
    L = [0.1 * v for v in n]
    def f(v):
        return f"{v:.10g}"
    return f"""
cell {f(L[0])} {f(L[1])} {f(L[2])}
max-layers {max_layers}
boxes 4
method {method}
save bench.vmg
begin layers
  sor {sor}
  tol {tol}
  v-steps {steps}
  mesh-size {n[0]} {n[1]} {n[2]}
  bc a periodic
  bc b periodic
  bc c- neumann
  bc c+ dirichlet
  interp full
end layers
begin box
  llc {f(0.4 * L[0])} {f(0.4 * L[1])} {f(0.05 * L[2])}
  cell {f(0.2 * L[0])} {f(0.2 * L[1])} {f(0.9 * L[2])}
  value 0.
  rho 3.
  constant false
end box
begin box
  llc {f(0.25 * L[0])} {f(0.25 * L[1])} 0.
  cell {f(0.5 * L[0])} {f(0.5 * L[1])} {f(0.05 * L[2])}
  value 0.5
end box
begin box
  llc {f(0.25 * L[0])} {f(0.25 * L[1])} {f(0.95 * L[2])}
  cell {f(0.5 * L[0])} {f(0.5 * L[1])} {f(0.05 * L[2])}
  value -0.5
end box
"""


def config5_input(n=(2048, 2048, 2048), tol="1.e-8", sor="1.7", steps=2) -> str:
    """2048^3 fully periodic case (config 5; config 2 is the same at 256^3): two plate electrodes +-0.5"""
    L = [0.1 * v for v in n]
    def f(v):
        return f"{v:.10g}"
    return f"""
cell {f(L[0])} {f(L[1])} {f(L[2])}
max-layers 8
boxes 4
method v-cycle
begin layers
  sor {sor}
  tol {tol}
  v-steps {steps}
  mesh-size {n[0]} {n[1]} {n[2]}
  bc all periodic
  interp full
end layers
begin box
  llc {f(0.25 * L[0])} {f(0.25 * L[1])} 0.
  cell {f(0.5 * L[0])} {f(0.5 * L[1])} {f(0.05 * L[2])}
  value 0.5
end box
begin box
  llc {f(0.25 * L[0])} {f(0.25 * L[1])} {f(0.95 * L[2])}
  cell {f(0.5 * L[0])} {f(0.5 * L[1])} {f(0.05 * L[2])}
  value -0.5
end box
"""


def config4_input(n=(256, 256, 4096), tol="1.e-8", sor="1.7", steps=2) -> str:
    """elongated electrode geometry (config 4): electrodes at the two z ends, `max-layers 4` keeps every level an exact
    halving, a per-layer block tightens the coarsest level"""
    L = [0.1 * v for v in n]
    def f(v):
        return f"{v:.10g}"
    return f"""
cell {f(L[0])} {f(L[1])} {f(L[2])}
max-layers 4
boxes 4
method v-cycle
begin layers
  sor {sor}
  tol {tol}
  v-steps {steps}
  mesh-size {n[0]} {n[1]} {n[2]}
  bc all periodic
  bc c- neumann
  bc c+ dirichlet
  interp full
end layers
begin layer -1
  tol 1.e-9
end layer
begin box
  llc {f(0.25 * L[0])} {f(0.25 * L[1])} 0.
  cell {f(0.5 * L[0])} {f(0.5 * L[1])} {f(0.05 * L[2])}
  value 0.5
end box
begin box
  llc {f(0.25 * L[0])} {f(0.25 * L[1])} {f(0.95 * L[2])}
  cell {f(0.5 * L[0])} {f(0.5 * L[1])} {f(0.05 * L[2])}
  value -0.5
end box
"""


def host_cores():
    """the cores this process may use: its affinity mask, capped by the thread count the environment prescribes"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    if cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline(seconds_target: float = 12.0, n=(512, 512, 512)):
    """The reference's gs_step on this host's cores, on a bounded sample of the bench geometry: every core this
    process may use, and one thread (the only reproducible figure: the reference's OpenMP sweep is racy)."""
    from multigrid_amd import mgio
    top = mgio.iomg_read(mgio.MgIO(text=config3_input(n)))
    cores = host_cores()
    visible = os.cpu_count() or 1
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver_omp_O3_f64")
    flags = "-O3 -fopenmp"
    if not os.path.exists(exe):
        exe, flags = os.path.join(ROOT, "oracle", "_ref", "ref_driver_omp_f64"), "-O2 -fopenmp"
    boxes = [b for b in top.box if not np.all(b.place == 0)]
    if os.path.exists(exe):
        try:
            def run(k, threads):
                with tempfile.TemporaryDirectory() as td:
                    case, out = os.path.join(td, "case.txt"), os.path.join(td, "out.bin")
                    with open(case, "w") as fcase:
                        fcase.write(f"5\n{n[0]} {n[1]} {n[2]}\n{' '.join(map(str, top.bc6()))}\n{top.sor!r}\n-1 -1 -1\n{len(boxes)}\n")
                        for b in boxes:
                            fcase.write(" ".join(str(int(v)) for v in b.place) + f" {b.val!r} {b.rho!r} {int(b.constant)}\n")
                        fcase.write(f"{k}\n1 1 1\n0\n")
                    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
                    txt = subprocess.run([exe, case, out], env=env, check=True, stdout=subprocess.PIPE,
                                         universal_newlines=True, timeout=600).stdout
                    return float(txt.split("REF_SWEEP_SECONDS")[1].split()[0])
            cells = n[0] * n[1] * n[2]
            t2 = run(2, cores)
            k = int(max(2, min(400, seconds_target / max(t2 / 2, 1e-6))))
            t = run(k, cores)
            k1 = int(max(1, min(k, seconds_target / max(t2 / 2 * cores * 0.5, 1e-6))))
            t1 = run(k1, 1)
            return dict(value=cells * k / t / 1e9, unit="GLUP/s", cores=cores, kind="reference",
                        sample=f"{k} gs_step sweeps (reference's lexicographic OpenMP smoother, amdflang {flags}, fp64 build) "
                               f"on {n[0]}x{n[1]}x{n[2]} of the bench geometry, {t:.1f} s on {cores} threads "
                               f"(host shows {visible} CPUs); 1 thread: {k1} sweeps in {t1:.1f} s",
                        one_thread_value=cells * k1 / t1 / 1e9, host_cpus_visible=visible)
        except Exception as e:   # binary not runnable here: fall through to the port
            sys.stderr.write(f"[bench] reference binary failed ({e}); timing the C oracle instead\n")
    from oracle import oracle as orc
    from tests.helpers import oracle_levels
    n = (256, 256, 256)
    top = mgio.iomg_read(mgio.MgIO(text=config3_input(n)))
    L = oracle_levels(top)[0]
    V = np.zeros(L.shape())
    orc.setup(L, V, True)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds_target:
        orc.sweep(L, V, orc.ORDER_RB)
        k += 1
    t = time.perf_counter() - t0
    return dict(value=n[0] * n[1] * n[2] * k / t / 1e9, unit="GLUP/s", cores=1, kind="port",
                sample=f"{k} red-black sweeps of the C oracle (1 thread) on {n[0]}x{n[1]}x{n[2]}, {t:.1f} s")


def vcycle_cpu_baseline(ctx, n=80, tol="1.e-8", sor="1.7"):
    """The first-named metric beside a CPU figure: the reference's own `mg` program (oracle/_ref/mg_ref_omp_O3_f64: its
    parser, hierarchy, cycle control and OpenMP smoother, real(8) build) solving the config-3 input `method v-cycle` to
    `tol` at n^3 — the size it finishes inside the bench's budget — on every core this process may use, and the same
    input file solved on the GPU.  The reference sweeps in its own (racy under OpenMP) order, the GPU in red-black
    order: sweep counts are reported for both."""
    import re
    from multigrid_amd import cycle, mgio
    text = config3_input((n, n, n), tol=tol, sor=sor)
    out = {"workload": f"config 3 at {n}x{n}x{n}, method v-cycle, tol {tol}, sor {sor}, v-steps 2"}
    exe = os.path.join(ROOT, "oracle", "_ref", "mg_ref_omp_O3_f64")
    cores = host_cores()
    if os.path.exists(exe):
        try:
            with tempfile.TemporaryDirectory() as td:
                with open(os.path.join(td, "mg.input"), "w") as f:
                    f.write(text)
                t0 = time.perf_counter()
                r = subprocess.run([exe, "mg.input"], cwd=td, env=dict(os.environ, OMP_NUM_THREADS=str(cores)),
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
                t = time.perf_counter() - t0
            done = re.findall(r"Completed:\s*(\d+):\s*(\d+) in\s+(\d+) itt", r.stdout)
            out["cpu"] = {"seconds": t, "cores": cores, "kind": "reference", "returncode": r.returncode,
                          "program": "the reference's mg (amdflang -O3 -fopenmp, real(8) build), whole program incl. input and .VMG output",
                          "sweeps_on_upper_level_per_pair": {f"{a}:{b}": int(k) for a, b, k in done}}
        except Exception as e:
            out["cpu"] = {"seconds": None, "error": str(e)}
    else:
        out["cpu"] = {"seconds": None, "error": "oracle/_ref/mg_ref_omp_O3_f64 is not built"}
    top = mgio.iomg_read(mgio.MgIO(text=text))
    solver = cycle.Solver(top, ctx=ctx, print_sums=False, max_sweeps=400000)
    try:
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status = "converged"
        try:
            solver.mg_gs_cds(cycle.CDS_W_CYCLE)
        except Exception as e:
            status = f"stopped: {e}"
        torch.cuda.synchronize()
        out["gpu"] = {"seconds": time.perf_counter() - t0, "status": status,
                      "sweeps_per_level": [int(g.itt) for g in top.levels()]}
    finally:
        solver.close()
    if out["cpu"].get("seconds") and out["gpu"]["status"] == "converged":
        out["speedup"] = out["cpu"]["seconds"] / out["gpu"]["seconds"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", dest="n", type=int, default=512, help="--gpus 1: cells per dimension of config 3")
    ap.add_argument("--config", type=int, default=0, help="--gpus N > 1: 5 (2048^3 periodic, default) or 4 (256 x 256 x 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vcycle", action="store_true")
    ap.add_argument("--no-fortran-host", action="store_true", help="skip the v-cycle solve through the Fortran host")
    ap.add_argument("--vcycle-tol", default="1.e-8")
    ap.add_argument("--sor", default="1.7")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 sweeps of the same level")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 two-pass, 2 fused (one sweep per pass), 3 fused with pair passes everywhere")
    ap.add_argument("--own-stream", action="store_true", help="let the library create its stream (events then see nothing)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--no-dist-vcycle", action="store_true",
                    help="with --gpus > 1: skip the v-cycle solve of config 3 with every level decomposed into z slabs")
    ap.add_argument("--dist-size", type=int, default=0, help="with --gpus > 1: cells per dimension instead of the configuration's (rehearsals)")
    ap.add_argument("--no-verify", action="store_true",
                    help="with --gpus > 1: skip the comparison of the reductions with the undivided level on rank 0")
    ap.add_argument("--no-overlap", action="store_true", help="exchange the halo planes after the whole sweep")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from multigrid_amd import cycle, lib, mgio

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.own_stream and args.gpus > 1:
        raise SystemExit("--own-stream cannot be combined with --gpus > 1: RCCL transfers are ordered on torch's current "
                         "stream, which the library must share")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local_rank %= max(1, torch.cuda.device_count())      # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # a stream of torch's own (the default stream has handle 0): the library enqueues on it, so the
    # events below bracket exactly the kernels of the timed region
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    ctx = lib.Context(local_rank, None if args.own_stream else tstream.cuda_stream)
    ctx.set_kernel(args.kernel)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    RED = lib.RED_ERR | lib.RED_SUM

    def progress(msg):
        """a line on stderr between the phases (stdout carries the one JSON line only)"""
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    def timed(fn, reps):
        """device time per call of fn on the library's stream (HIP events) and host wall time per call"""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / reps, (time.perf_counter() - t0) / reps * 1e3

    if world == 1:
        n = args.n
        dims = (n, n, n)
        top = mgio.iomg_read(mgio.MgIO(text=config3_input(dims, tol=args.vcycle_tol, sor=args.sor)))
        solver = cycle.Solver(top, ctx=ctx)
        solver.bring_back(top)
        solver.setup(top, True)
        lv = top.dev

        def step():
            lv.sweep(1, RED)
        fetch = lv.result
        local_cells = total_cells = n ** 3
        workload = (f"config 3: {n}x{n}x{n} mixed periodic/Neumann/Dirichlet, 2 plate electrodes + rho=3 block, "
                    f"finest level, one red-black SOR sweep with fused err/sum reductions per step")
        scaling = "strong"      # (one GPU: nothing scales; the multi-GPU runs of this bench keep the total work fixed)
    else:
        from multigrid_amd import slab
        cfg = args.config or 5
        if cfg == 5:
            dims = (2048, 2048, 2048) if not args.dist_size else (args.dist_size,) * 3
            text = config5_input(dims, tol=args.vcycle_tol, sor=args.sor)
            name = f"config 5: {dims[0]}x{dims[1]}x{dims[2]} fully periodic, 2 plate electrodes"
        elif cfg == 4:
            dims = (256, 256, 4096) if not args.dist_size else (args.dist_size, args.dist_size, 16 * args.dist_size)
            text = config4_input(dims, tol=args.vcycle_tol, sor=args.sor)
            name = f"config 4: {dims[0]}x{dims[1]}x{dims[2]} elongated, electrodes at the z ends"
        else:
            raise SystemExit("--config 4 or 5")
        top = mgio.iomg_read(mgio.MgIO(text=text))
        sl = slab.SlabLevel(ctx, top, rank, world, overlap=not args.no_overlap)
        sl.setup(True)
        step = sl.sweep
        fetch = sl.result
        total_cells = dims[0] * dims[1] * dims[2]
        local_cells = dims[0] * dims[1] * sl.n3
        workload = (f"{name}, finest level in {world} z slabs, one red-black SOR sweep with fused err/sum reductions "
                    f"and the halo exchange per step")
        scaling = "strong"

    progress(f"levels set up; {args.warmup} warm-up and {args.steps} timed steps")
    for _ in range(args.warmup):
        step()
    fetch()
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    e2, s, _ = fetch()
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    wall = float(dt.item())
    ms_per_step = wall / args.steps * 1e3
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # device time of one sweep on the kernel's stream
    glups = total_cells / (wall / args.steps) / 1e9
    achieved = BYTES_PER_LUP * local_cells / (kernel_ms * 1e-3) / 1e9   # per GPU

    out = {
        "metric": METRIC, "value": glups, "unit": "GLUP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "parallelism": "single GPU" if world == 1 else f"z-slab x{world}",
                   "sor": float(args.sor)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel_ms": kernel_ms, "bytes_per_lup": BYTES_PER_LUP},
        "hbm_peak_pct": 100.0 * achieved / HBM_PEAK_GBS,
        "check": {"err_sq_max": e2, "abs_sum": s},
    }
    if world > 1 and not args.no_verify:
        progress("steps timed; rank 0 repeats them on the undivided level")
        # rank 0 repeats the same sweeps on the undivided level (it fits one GPU) and compares the reductions of the
        # last sweep bit for bit: red-black results and the exact sum do not depend on the decomposition
        if rank == 0:
            try:
                from multigrid_amd import grid as G
                whole = ctx.level(top.layer, [int(v) for v in top.n], top.dtype)
                whole.set_params(top.sor, [float(v) for v in top.a], top.bc6())
                whole.set_boxes(*G.box_table(top))
                whole.bring_back()
                whole.setup(True)
                for _ in range(args.warmup + args.steps):
                    whole.sweep(1, RED)
                we2, ws, _ = whole.result()
                whole.close()
                out["check"]["undivided"] = {"err_sq_max": we2, "abs_sum": ws, "identical": bool(we2 == e2 and ws == s)}
            except Exception as e:        # the measurement stands; the comparison is reported as not done
                out["check"]["undivided"] = {"error": str(e)}
        barrier()
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr) and world == 1:
        try:
            out["roofline"]["traffic"] = json.load(open(tr)).get("hbm_bytes_per_launch")
        except Exception:
            pass

    if world == 1:
        # two sweeps in one pass over the field (mgx_level_sweep(lv, 2, ...): the form gs_V's legs take), reductions on
        # the second sweep as the legs ask for them; 16 B of algorithmic traffic per cell per PASS
        def pair():
            lv.sweep(2, lib.RED_ERR)
        for _ in range(3):
            pair()
        pk, _ = timed(pair, max(5, args.steps // 2))
        out["roofline"]["pair_pass"] = {"kernel_ms": pk, "sweeps_per_pass": 2, "bytes_per_cell_per_pass": BYTES_PER_LUP,
                                        "achieved": BYTES_PER_LUP * local_cells / (pk * 1e-3) / 1e9,
                                        "frac": BYTES_PER_LUP * local_cells / (pk * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "glups": 2 * local_cells / (pk * 1e-3) / 1e9}

    if world == 1 and not args.no_fp32:
        # the same level in real(4), the precision the reference ships (t_mg.f90:8): single sweeps with both reductions
        # and passes of two sweeps with the error reduction; 8 B of algorithmic traffic per update, resp. per cell and pass
        progress("the same sweeps in fp32")
        try:
            top32 = mgio.iomg_read(mgio.MgIO(text=config3_input(dims, tol=args.vcycle_tol, sor=args.sor)), dtype=np.float32)
            s32 = cycle.Solver(top32, ctx=ctx)
            s32.bring_back(top32)
            s32.setup(top32, True)
            l32 = top32.dev
            for _ in range(3):
                l32.sweep(1, RED)
            k1, _ = timed(lambda: l32.sweep(1, RED), max(10, args.steps // 2))
            for _ in range(2):
                l32.sweep(2, lib.RED_ERR)
            k2, _ = timed(lambda: l32.sweep(2, lib.RED_ERR), max(5, args.steps // 4))
            b32 = 8.0
            out["fp32"] = {"sweep": {"kernel_ms": k1, "glups": local_cells / (k1 * 1e-3) / 1e9, "bytes_per_lup": b32,
                                     "frac": b32 * local_cells / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "pair_pass": {"kernel_ms": k2, "glups": 2 * local_cells / (k2 * 1e-3) / 1e9,
                                         "bytes_per_cell_per_pass": b32,
                                         "frac": b32 * local_cells / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            s32.close()
        except Exception as e:            # the fp64 measurement stands
            out["fp32"] = {"error": str(e)}

    if world == 1:
        # the same sweep in the reference's own order (MGX_KERNEL_LEX: results of the reference on one thread, bit for
        # bit), for the record: it is the slow, order-faithful mode, not the metric
        ctx.set_kernel(lib.KERNEL_LEX)
        try:
            step()
            lk, _ = timed(step, 5)
            out["reference_order"] = {"kernel_ms": lk, "glups": local_cells / (lk * 1e-3) / 1e9,
                                      "note": "gs in the reference's sweep order (all tiles in one launch, each waiting for the tiles before it), both reductions"}
        finally:
            ctx.set_kernel(args.kernel)

    if world == 1 and not args.no_vcycle:
        progress("steps timed; v-cycle solve of the same input")
        # complete `method v-cycle` solve of the same input (fresh state), timed to its own tolerance
        for g in top.levels():
            g.itt = 0
        solver.history = []
        solver.total_sweeps = 0
        solver.print_sums = False     # the sums gs_V evaluates only to print them (m_gs_CDS.F90:238,245,269,275)
        torch.cuda.synchronize()
        tv0 = time.perf_counter()
        status = "converged"
        try:
            solver.max_sweeps = 200000
            solver.mg_gs_cds(cycle.CDS_W_CYCLE)
        except Exception as e:
            status = f"stopped: {e}"
        torch.cuda.synchronize()
        tv = time.perf_counter() - tv0
        h = solver.history_array()
        out["vcycle"] = {"seconds": tv, "status": status, "tol": float(args.vcycle_tol),
                         "sweeps_per_level": [int(g.itt) for g in top.levels()],
                         "levels": [[int(v) for v in g.n] for g in top.levels()],
                         "final_err": float(h["err"][-1]) if len(h) else None,
                         "lattice_updates": int(sum(int(g.itt) * int(np.prod(g.n)) for g in top.levels()))}
        if status == "converged":
            # the timed steps above start from the set-up state (most of the field still zero); the same launches on the
            # converged potential: sweep time depends on the values (bit activity), not on the path
            for _ in range(3):
                step()
            sk, sw = timed(step, 20)
            out["vcycle"]["ms_per_step_on_solution"] = sw
            ach = BYTES_PER_LUP * local_cells / (sk * 1e-3) / 1e9
            out["roofline"]["on_solution"] = {"kernel_ms": sk, "achieved": ach, "frac": ach / HBM_PEAK_GBS}
            for _ in range(3):
                pair()
            pk, _ = timed(pair, 10)
            out["roofline"]["pair_pass"]["on_solution"] = {"kernel_ms": pk, "glups": 2 * local_cells / (pk * 1e-3) / 1e9,
                                                           "frac": BYTES_PER_LUP * local_cells / (pk * 1e-3) / 1e9 / HBM_PEAK_GBS}
    exe = os.path.join(ROOT, "multigrid_amd", "fortran", "mg_gpu_f64")
    if world == 1 and not args.no_vcycle and not args.no_fortran_host and os.path.exists(exe):
        progress("v-cycle solve done; the same solve through the Fortran host")
        # the same solve driven by the Fortran host (the reference's cycle control over the C ABI, ISO_C_BINDING), as
        # a child process with a context of its own
        with tempfile.TemporaryDirectory() as td:
            case = os.path.join(td, "case.txt")
            mgio.write_host_case(top, cycle.CDS_W_CYCLE, case)
            try:
                r = subprocess.run([exe, case, "-", "noprint"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   universal_newlines=True, timeout=600)
                sec = [float(l.split()[-1]) for l in r.stdout.splitlines() if l.startswith("mg_gpu: solve seconds")]
                per = [[int(v) for v in l.split(":")[-1].split()] for l in r.stdout.splitlines() if "sweeps per level" in l]
                out["vcycle_fortran_host"] = {"seconds": sec[0] if sec else None, "returncode": r.returncode,
                                              "sweeps_per_level": per[0] if per else None}
            except Exception as e:
                out["vcycle_fortran_host"] = {"seconds": None, "error": str(e)}
    if world > 1 and not args.no_dist_vcycle:
        progress("v-cycle solve of config 3 with every level decomposed over the ranks")
        # the headline solve (config 3, tol 1e-8) with every level of the hierarchy as z slabs aligned between levels over
        # the N GPUs: transfers are local, each is followed by one halo refresh, reductions are all-reduced (exact
        # sum: identical on every rank), thin coarse levels are held whole by every rank
        sl.lv.close()
        n3 = args.n if not args.dist_size else args.dist_size
        vtop = mgio.iomg_read(mgio.MgIO(text=config3_input((n3, n3, n3), tol=args.vcycle_tol, sor=args.sor)))
        status = "converged"
        try:
            dsolver = slab.SlabSolver.build(vtop, ctx, rank, world, overlap=not args.no_overlap, print_sums=False,
                                            max_sweeps=200000)
            barrier()
            tv0 = time.perf_counter()
            dsolver.mg_gs_cds(cycle.CDS_W_CYCLE)
            barrier()
            tv = time.perf_counter() - tv0
            h = dsolver.history_array()
            out["vcycle"] = {"seconds": tv, "status": status, "tol": float(args.vcycle_tol),
                             "workload": f"config 3: {n3}^3 over {world} z slabs",
                             "sweeps_per_level": [int(g.itt) for g in vtop.levels()],
                             "levels": [[int(v) for v in g.n] for g in vtop.levels()],
                             "final_err": float(h["err"][-1]) if len(h) else None,
                             "lattice_updates": int(sum(int(g.itt) * int(np.prod(g.n)) for g in vtop.levels()))}
        except Exception as e:
            out["vcycle"] = {"seconds": None, "status": f"stopped: {e}", "tol": float(args.vcycle_tol)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress("CPU baseline: the reference's smoother on the host cores")
        out["cpu_baseline"] = cpu_baseline()
        if not args.no_vcycle:
            progress("CPU baseline of the V-cycle: the reference's mg program on a size it finishes in the budget")
            out["vcycle_cpu_baseline"] = vcycle_cpu_baseline(ctx)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
