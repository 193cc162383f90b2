"""The multi-GPU form of bench.py, rehearsed: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
as a fresh child process, two ranks on the one GPU of the box, planes through gloo (RCCL refuses two ranks on one
device).  Guards the Python-level path the scaling runs take: argument handling, slab set-up, the timed loop with its
barriers, the comparison with the undivided level and the decomposed V-cycle solve."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--vcycle-tol", "1.e-5"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [["--dist-size", "64"], ["--config", "4", "--dist-size", "48"]],
                         ids=["config5_64", "config4_48x48x768"])
def test_two_rank_bench_line(extra):
    out = _run_bench(extra)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["unit"] == "GLUP/s" and out["value"] > 0
    assert out["check"]["undivided"]["identical"] is True
    assert out["vcycle"]["status"] == "converged" and out["vcycle"]["seconds"] > 0
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
