#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generate tests/golden/*.npz from the reference itself.

Runs the binaries `make -C oracle ref` built from /root/reference/src
(oracle/_ref/ref_driver_{f32,f64}: our driver over the reference's modules;
oracle/_ref/mg_ref_{f32,f64}: the reference's own `mg` program), always with
one thread, and stores inputs + outputs as small fixtures.  Only data is
stored — no reference source.  Run in the build container only (the GPU box
has no reference and never regenerates fixtures).

    python oracle/make_golden.py            # all fixtures
    python oracle/make_golden.py sweeps     # one group
"""
from __future__ import annotations

import itertools
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ENV = dict(os.environ, OMP_NUM_THREADS="1")
DT = {"f32": np.float32, "f64": np.float64}


def run_driver(prec, mode, n, bc, sor, a, boxes, k, nc, seed):
    with tempfile.TemporaryDirectory() as td:
        case = os.path.join(td, "case.txt")
        out = os.path.join(td, "out.bin")
        with open(case, "w") as f:
            f.write(f"{mode}\n{n[0]} {n[1]} {n[2]}\n{' '.join(map(str, bc))}\n{sor!r}\n")
            f.write(" ".join(repr(float(x)) for x in a) + "\n")
            f.write(f"{len(boxes)}\n")
            for b in boxes:
                f.write(" ".join(map(str, b[0])) + f" {b[1]!r} {b[2]!r} {int(b[3])}\n")
            f.write(f"{k}\n{nc[0]} {nc[1]} {nc[2]}\n{seed}\n")
        subprocess.check_call([os.path.join(REF, f"ref_driver_{prec}"), case, out], env=ENV)
        return open(out, "rb").read()


def take(buf, off, dtype, count):
    a = np.frombuffer(buf, dtype=dtype, count=count, offset=off)
    return a, off + a.nbytes


def box_arrays(boxes):
    place = np.array([b[0] for b in boxes], dtype=np.int32).reshape(len(boxes), 6)
    val = np.array([b[1] for b in boxes], dtype=np.float64)
    rho = np.array([b[2] for b in boxes], dtype=np.float64)
    const = np.array([b[3] for b in boxes], dtype=np.int32)
    return dict(box_place=place, box_val=val, box_rho=rho, box_const=const)


SWEEP_CASES = {
    # name: (n, bc, sor, a, boxes[(place,val,rho,const)], k, seed)
    "sweep_24_mixed": ((24, 24, 24), (1, 1, 2, 3, 3, 2), 1.8, (-1, -1, -1),
                       [((3, 8, 4, 20, 2, 3), 0.5, 1.0, 1), ((15, 22, 4, 20, 21, 23), -0.5, 1.0, 1),
                        ((10, 14, 9, 15, 4, 20), 0.0, 3.0, 0)], 3, 12345),
    "sweep_33x20x17_w": ((33, 20, 17), (3, 2, 1, 1, 2, 3), 1.5, (0.5, 0.3, 0.2),
                         [((1, 5, 1, 20, 1, 2), 0.3, 1.0, 1), ((4, 12, 6, 9, 2, 9), 0.7, 2.5, 0),
                          ((10, 30, 8, 12, 5, 16), -0.25, 1.0, 1), ((0, 0, 0, 0, 0, 0), 0.0, 1.0, 0)], 4, 0),
    "sweep_16_zero_start": ((16, 16, 16), (2, 2, 2, 2, 2, 2), 1.8, (-1, -1, -1),
                            [((4, 12, 4, 12, 1, 1), 0.5, 1.0, 1), ((4, 12, 4, 12, 16, 16), -0.5, 1.0, 1)], 5, 0),
    "sweep_12x10x8_periodic": ((12, 10, 8), (1, 1, 1, 1, 1, 1), 1.2, (-1, -1, -1),
                               [((2, 3, 2, 9, 2, 7), 1.0, 1.0, 1), ((8, 11, 3, 4, 3, 5), 0.1, 4.0, 0)], 3, 777),
}


def gen_sweeps():
    for name, (n, bc, sor, a, boxes, k, seed) in SWEEP_CASES.items():
        for prec, dt in DT.items():
            buf = run_driver(prec, 1, n, bc, sor, a, boxes, k, (1, 1, 1), seed)
            vol = (n[0] + 2) * (n[1] + 2) * (n[2] + 2)
            shp = (n[2] + 2, n[1] + 2, n[0] + 2)
            off = 0
            v0, off = take(buf, off, dt, vol)
            cnt, off = take(buf, off, np.int32, 1)
            st, off = take(buf, off, np.float64, 2)
            vs, errs, sums = [], [], []
            for _ in range(k):
                v, off = take(buf, off, dt, vol)
                es, off = take(buf, off, np.float64, 2)
                vs.append(v.reshape(shp))
                errs.append(es[0])
                sums.append(es[1])
            assert off == len(buf)
            np.savez_compressed(
                os.path.join(OUT, f"{name}_{prec}.npz"), n=np.array(n, np.int32), bc=np.array(bc, np.int32),
                sor=np.float64(sor), a=np.array(a, np.float64), k=np.int32(k), V0=v0.reshape(shp),
                count=cnt[0], sum0=st[0], tolerance=st[1], V=np.stack(vs), err=np.array(errs),
                sum=np.array(sums), **box_arrays(boxes))
            print("wrote", name, prec)


TRANSFER_CASES = {
    # name: (n parent, n child, bc, parent boxes, seed)
    "xfer_32_16": ((32, 32, 32), (16, 16, 16), (2, 2, 2, 2, 2, 2), [((5, 9, 5, 9, 5, 9), 0.5, 1.0, 1)], 4242),
    "xfer_24x16x40": ((24, 16, 40), (12, 8, 20), (1, 1, 3, 2, 2, 3), [((2, 4, 2, 15, 3, 38), -0.5, 1.0, 1)], 99),
    "xfer_odd_30x19x30": ((30, 19, 30), (15, 9, 15), (2, 3, 1, 1, 3, 2), [], 31337),
    "xfer_nonhalved_32x32x64": ((32, 32, 64), (32, 32, 32), (2, 2, 2, 2, 1, 1), [], 5),
}


def gen_transfer():
    for name, (n, nc, bc, boxes, seed) in TRANSFER_CASES.items():
        for prec, dt in DT.items():
            for mname, method in (("full", 1), ("half", 2)):
                volp = (n[0] + 2) * (n[1] + 2) * (n[2] + 2)
                volc = (nc[0] + 2) * (nc[1] + 2) * (nc[2] + 2)
                shp = (n[2] + 2, n[1] + 2, n[0] + 2)
                shc = (nc[2] + 2, nc[1] + 2, nc[0] + 2)
                buf = run_driver(prec, 2, n, bc, 1.8, (-1, -1, -1), boxes, method, nc, seed)
                vp, off = take(buf, 0, dt, volp)
                vc, off = take(buf, off, dt, volc)
                assert off == len(buf)
                buf = run_driver(prec, 3, n, bc, 1.8, (-1, -1, -1), boxes, method, nc, seed)
                pc, off = take(buf, 0, dt, volc)
                pp, off = take(buf, off, dt, volp)
                assert off == len(buf)
                np.savez_compressed(
                    os.path.join(OUT, f"{name}_{mname}_{prec}.npz"), n=np.array(n, np.int32),
                    nc=np.array(nc, np.int32), bc=np.array(bc, np.int32), method=np.int32(method),
                    res_parent=vp.reshape(shp), res_child=vc.reshape(shc),
                    pro_child=pc.reshape(shc), pro_parent=pp.reshape(shp), **box_arrays(boxes))
                print("wrote", name, mname, prec)


def gen_setup():
    n = (6, 5, 4)
    boxes = [((2, 3, 1, 2, 1, 4), 0.75, 1.0, 1), ((3, 6, 2, 5, 2, 2), -1.5, 2.0, 0), ((5, 6, 4, 5, 3, 4), 0.25, 1.0, 1)]
    shp = (n[2] + 2, n[1] + 2, n[0] + 2)
    vol = int(np.prod(shp))
    for prec, dt in DT.items():
        bcs, outs = [], []
        for bc in itertools.product((1, 2, 3), repeat=6):
            buf = run_driver(prec, 4, n, bc, 1.8, (-1, -1, -1), boxes, 0, (1, 1, 1), 2024)
            v, off = take(buf, 0, dt, vol)
            assert off == len(buf)
            bcs.append(bc)
            outs.append(v.reshape(shp))
        # the pre-setup random field is the same for every bc: regenerate it with the same LCG
        s, raw = 2024, np.empty(vol, dtype=dt)
        for i in range(vol):
            s = (s * 1103515245 + 12345) % 2147483648
            raw[i] = dt(s // 128) / dt(16777216.0) - dt(0.5)
        np.savez_compressed(os.path.join(OUT, f"setup_6x5x4_{prec}.npz"), n=np.array(n, np.int32),
                            bc=np.array(bcs, np.int32), V_in=raw.reshape(shp), V_out=np.stack(outs),
                            **box_arrays(boxes))
        print("wrote setup", prec)


# ---------------------------------------------------------------------------------------------
# whole solves through the reference's own program (parser + cycle control + writer)

SOLVE_INPUTS = {
    "solve_64_bu": """
cell 6.4 6.4 6.4
max-layers 5
boxes 3
save out.vmg
begin layers
  sor 1.8
  tol 1.e-5
  v-steps 2
  mesh-size 64 64 64
  bc all dirichlet
  bc a periodic
  interp full
end layers
begin box
  llc 1.6 1.6 0.
  cell 3.2 3.2 0.32
  value 0.5
end box
begin box
  llc 1.6 1.6 6.08
  cell 3.2 3.2 0.32
  value -0.5
end box
""",
    "solve_80_bu": """
cell 8. 8. 8.
max-layers 5
boxes 4
save out.vmg
begin layers
  sor 1.8
  tol 1.e-4
  v-steps 2
  mesh-size 80 80 80
  bc all dirichlet
  bc c- neumann
  interp full
end layers
begin layer -1
  tol 1.e-6
end layer
begin box
  llc 2. 2. 0.
  cell 4. 4. 0.4
  value 0.5
end box
begin box
  llc 2. 2. 7.6
  cell 4. 4. 0.4
  value -0.5
end box
begin box
  llc 3.2 3.2 0.4
  cell 1.6 1.6 7.2
  value 0.
  rho 3.
  constant false
end box
""",
    "solve_80_v": """
cell 8. 8. 8.
max-layers 5
boxes 4
method v-cycle
save out.vmg
begin layers
  sor 1.8
  tol 1.e-4
  v-steps 2
  mesh-size 80 80 80
  bc all periodic
  interp full
end layers
begin box
  llc 2. 2. 0.
  cell 4. 4. 0.4
  value 0.5
end box
begin box
  llc 2. 2. 7.6
  cell 4. 4. 0.4
  value -0.5
end box
""",
    "solve_64x48x80_v_half": """
cell 6.4 4.8 8.
max-layers 5
boxes 2
method w
save out.vmg
begin layers
  sor 1.7
  tol 1.e-4
  v-steps 3
  mesh-size 64 48 80
  bc all dirichlet
  bc b neumann
  interp half
end layers
begin box
  llc 0. 1. 2.
  cell 0.4 2.8 4.
  value 1.
end box
begin box
  llc 6. 1. 2.
  cell 0.4 2.8 4.
  value -1.
end box
""",
}


# grid_add_box (t_mg.f90:323-435) on orthogonal and skewed cells: levels, cell matrices (lattice vectors as columns),
# offsets and boxes (llc, box_cell) incl. boxes outside the cell, mirrored boxes (negative extents), points and lines
def _boxplace_cases():
    rng = np.random.default_rng(20240611)
    cases = {}
    ortho = np.diag([2.4, 1.5, 3.1])
    skew = np.array([[2.4, 0.6, 0.3], [0.2, 1.5, -0.4], [0.1, 0.25, 3.1]])      # columns = lattice vectors
    hexa = np.array([[3.0, -1.5, 0.0], [0.0, 2.598076211353316, 0.0], [0.0, 0.0, 4.0]])
    for name, n, cell, off in (("ortho", (24, 15, 31), ortho, (0., 0., 0.)), ("skew", (24, 15, 31), skew, (0.1, -0.2, 0.05)),
                               ("hexa", (30, 30, 16), hexa, (0., 0., 0.)), ("thin", (70, 3, 5), skew, (0., 0., 0.))):
        boxes = []
        ext = np.abs(cell).sum(axis=1)
        lo = np.minimum(cell, 0).sum(axis=1) + np.array(off)
        for _ in range(10):
            llc = lo + rng.uniform(-0.1, 0.9, 3) * ext
            bc = np.diag(rng.uniform(0.02, 0.6, 3) * ext)
            boxes.append((llc, bc))
        llc = lo + 0.8 * ext
        boxes.append((llc, np.diag(-0.5 * ext)))                   # mirrored: urc below llc (t_mg.f90:378-385)
        boxes.append((lo + 0.5 * ext, np.zeros((3, 3))))          # a point (grid_add_point, :437-448)
        bl = np.zeros((3, 3)); bl[1, 1] = 0.4 * ext[1]
        boxes.append((lo + 0.3 * ext, bl))                         # a line (grid_add_line, :450-463)
        boxes.append((lo + 5.0 * ext, np.diag(0.1 * ext)))         # outside the cell
        sk = np.array([[0.3, 0.1, 0.0], [0.0, 0.25, 0.05], [0.02, 0.0, 0.4]]) * ext[:, None]
        boxes.append((lo + 0.2 * ext, sk))                         # skewed box_cell: only its row sums enter (:360)
        boxes.append((lo - 0.01 * ext, np.diag(1.2 * ext)))        # the whole cell
        cases[name] = (n, cell, np.array(off), boxes)
    return cases


def gen_boxplace():
    out = {}
    for name, (n, cell, off, boxes) in _boxplace_cases().items():
        with tempfile.TemporaryDirectory() as td:
            case = os.path.join(td, "case.txt")
            with open(case, "w") as f:
                f.write(f"{n[0]} {n[1]} {n[2]}\n")
                f.write(" ".join(repr(float(v)) for v in cell.ravel(order="F")) + "\n")
                f.write(" ".join(repr(float(v)) for v in off) + "\n")
                f.write(f"{len(boxes)}\n")
                for llc, bc in boxes:
                    f.write(" ".join(repr(float(v)) for v in llc) + " " +
                            " ".join(repr(float(v)) for v in bc.ravel(order="F")) + "\n")
            txt = subprocess.check_output([os.path.join(REF, "ref_boxplace_f64"), case], env=ENV).decode()
        place = np.array([[int(t) for t in line.split()] for line in txt.strip().splitlines()], dtype=np.int64)
        assert place.shape == (len(boxes), 6)
        out[f"{name}_n"] = np.array(n, np.int32)
        out[f"{name}_cell"] = cell
        out[f"{name}_offset"] = off
        out[f"{name}_llc"] = np.array([b[0] for b in boxes])
        out[f"{name}_box_cell"] = np.array([b[1] for b in boxes])
        out[f"{name}_place"] = place
        print(name, place.tolist())
    np.savez_compressed(os.path.join(OUT, "box_place.npz"), **out)


def read_vmg(path, dt):
    """Fortran sequential unformatted (4-byte markers): comment(200), n(3) i4, offset(3) f8, cell(3,3) f8,
    then n3 records of (n1+2)(n2+2) reals: V(0:n1+1,0:n2+1,z), z=1..n3  (m_mg_save.F90:231-267)"""
    raw = open(path, "rb").read()
    off = 0

    def rec():
        nonlocal off
        ln = int(np.frombuffer(raw, "<i4", 1, off)[0])
        data = raw[off + 4: off + 4 + ln]
        assert int(np.frombuffer(raw, "<i4", 1, off + 4 + ln)[0]) == ln
        off += 8 + ln
        return data
    rec()
    n = np.frombuffer(rec(), "<i4")
    offset = np.frombuffer(rec(), "<f8")
    cell = np.frombuffer(rec(), "<f8").reshape(3, 3)
    planes = [np.frombuffer(rec(), dt).reshape(n[1] + 2, n[0] + 2) for _ in range(n[2])]
    assert off == len(raw)
    return n, offset, cell, np.stack(planes)


BOX_RE = re.compile(r"\+\+ Box \{ value:\s*(\S+) constant: ([TF]) place:((?:\s+-?\d+){6}) \}")


def parse_stdout(txt):
    """layer sizes, per-layer boxes as printed by print_grid (t_mg.f90:870-949), sweep counts, 7-decimal history"""
    layers, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"-- Layer: (\d+) size: (\d+) (\d+) (\d+)", line)
        if m:
            cur = dict(layer=int(m.group(1)), n=[int(m.group(i)) for i in (2, 3, 4)], boxes=[])
            layers.append(cur)
            continue
        m = re.search(r"-- tolerance:\s*(\S+)", line)
        if m and cur is not None:
            cur["tolerance_print"] = float(m.group(1))
        m = re.search(r"-- SOR: (\S+)", line)
        if m and cur is not None:
            cur["sor_print"] = float(m.group(1))
        m = BOX_RE.search(line)
        if m and cur is not None:
            cur["boxes"].append((float(m.group(1)), m.group(2) == "T", [int(v) for v in m.group(3).split()]))
    counts = [(int(a), int(b)) for a, b in re.findall(r"Completed \((\d+)\) cycle in (\d+) cycles", txt)]
    wcounts = [(int(a), int(b), int(c)) for a, b, c in re.findall(r"Completed: (\d+):(\d+) in (\d+) itt", txt)]
    hist = [(int(a), float(b), float(c), float(d)) for a, b, c, d in
            re.findall(r"^ itt\s+(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s*$", txt, flags=re.M)]
    return layers, counts, wcounts, hist


def run_mg(prec, text):
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "mg.input"), "w") as f:
            f.write(text)
        txt = subprocess.run([os.path.join(REF, f"mg_ref_{prec}"), "mg.input"], cwd=td, env=ENV, check=True,
                             stdout=subprocess.PIPE, universal_newlines=True).stdout
        n, offset, cell, V = read_vmg(os.path.join(td, "out.VMG"), DT[prec])
    return txt, n, offset, cell, V


def pack_solve(name, prec, text, txt, n, offset, cell, V, subsample=1):
    layers, counts, wcounts, hist = parse_stdout(txt)
    nb = max(len(l["boxes"]) for l in layers)
    place = np.zeros((len(layers), nb, 6), np.int32)
    const = np.zeros((len(layers), nb), np.int32)
    valp = np.zeros((len(layers), nb), np.float64)
    for i, l in enumerate(layers):
        for j, (v, c, p) in enumerate(l["boxes"]):
            place[i, j] = p
            const[i, j] = c
            valp[i, j] = v
    np.savez_compressed(
        os.path.join(OUT, f"{name}_{prec}.npz"), input=np.array(text), n=n, offset=offset, cell=cell,
        V=V[::subsample, ::subsample, ::subsample], subsample=np.int32(subsample),
        V_abs_sum=np.float64(np.abs(V.astype(np.float64)).sum()),
        layer_n=np.array([l["n"] for l in layers], np.int32), box_place=place, box_const=const, box_val_print=valp,
        tolerance_print=np.array([l["tolerance_print"] for l in layers]),
        bu_counts=np.array(counts, np.int32).reshape(-1, 2), w_counts=np.array(wcounts, np.int32).reshape(-1, 3),
        hist=np.array(hist, np.float64).reshape(-1, 4))
    print("wrote", name, prec, "counts", counts, wcounts)


def gen_solves():
    for name, text in SOLVE_INPUTS.items():
        for prec in DT:
            pack_solve(name, prec, text, *run_mg(prec, text))


def gen_test6():
    text = open("/root/reference/test/test6.input").read().replace("save test6_full.cube", "save out.vmg")
    for prec in DT:
        pack_solve("test6", prec, text, *run_mg(prec, text), subsample=6)


GROUPS = dict(sweeps=gen_sweeps, transfer=gen_transfer, setup=gen_setup, solves=gen_solves, test6=gen_test6,
              boxplace=gen_boxplace)

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for g in (sys.argv[1:] or list(GROUPS)):
        GROUPS[g]()
