"""GPU: whole solves with every level decomposed into z slabs over two ranks (both on this GPU; the planes travel
through the gloo backend here, through RCCL on a multi-GPU node) against the oracle's red-black solve of the
undivided hierarchy: residual history, sweep counts and the top-level potential, bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multigrid_amd import mgio                                   # noqa: E402
from oracle import oracle as orc                                 # noqa: E402
from tests.helpers import DT, config4_input, load, oracle_levels  # noqa: E402

pytestmark = pytest.mark.gpu

# the last case is the elongated multi-GPU configuration scaled down (x and y stop halving, four levels)
CASES = [("solve_64_bu", "f64", None, 4), ("solve_64x48x80_v_half", "f32", None, 4),
         ("solve_80_v", "f64", ("sor 1.8", "sor 1.6"), 4), ("config4", "f64", None, 4),
         # coarse levels replicated on both ranks (slabs thinner than the last number): two of four, one of three
         ("config4", "f64", None, 100), ("solve_80_v", "f64", ("sor 1.8", "sor 1.6"), 12), ("solve_64_bu", "f32", None, 20)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# V cycles with two sweeps per pass and ONE exchange of four planes a side on every slab level (MGX_PAIR_PASS=1 makes the
# library offer them at these sizes; by itself it does so from 12 million cells a slab on)
PAIR_CASES = [("solve_64x48x80_v_half", "f32", None, 4), ("solve_80_v", "f64", ("sor 1.8", "sor 1.6"), 4),
              ("config4", "f64", None, 4), ("solve_80_v", "f64", ("sor 1.8", "sor 1.6"), 12)]


@pytest.mark.parametrize("name,prec,sub,below", PAIR_CASES)
def test_two_rank_solve_with_pair_passes_matches_oracle(name, prec, sub, below, tmp_path):
    test_two_rank_solve_matches_oracle(name, prec, sub, below, tmp_path, pairs=True)


@pytest.mark.parametrize("name,prec,sub,below", CASES)
def test_two_rank_solve_matches_oracle(name, prec, sub, below, tmp_path, pairs=False):
    text = config4_input() if name == "config4" else str(load(f"{name}_{prec}")["input"])
    if sub:
        assert sub[0] in text
        text = text.replace(*sub)
    IO = mgio.MgIO(text=text)
    top = mgio.iomg_read(IO, dtype=DT[prec])
    method = mgio.read_method(IO)
    V, hist, per = orc.solve(oracle_levels(top), method, DT[prec], order=orc.ORDER_RB, sum_mode=orc.SUM_EXACT)
    case, out = str(tmp_path / "case.npz"), str(tmp_path / "out.npz")
    np.savez(case, input=text, dtype=np.dtype(DT[prec]).name)
    env = dict(os.environ, PYTHONPATH=ROOT)
    if pairs:
        env["MGX_PAIR_PASS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "dist_solve_worker.py"), case, out, "gloo", str(below)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=150)
    assert r.returncode == 0, r.stdout[-3000:]
    d = np.load(out)
    assert (4 in list(d["depths"])) == pairs and (int(d["depths"][0]) == 4) == pairs
    if below > 4:
        assert "ReplicatedDevice" in list(d["kinds"]) and str(d["kinds"][0]) == "SlabDevice"
    assert [int(v) for v in d["per"]] == per
    assert len(d["h_itt"]) == len(hist)
    for f in ("layer", "kind", "itt", "tol", "sum", "err"):
        assert np.array_equal(d[f"h_{f}"], hist[f]), f
    assert np.array_equal(d["planes"].view(np.uint8), V[1:-1].view(np.uint8))
