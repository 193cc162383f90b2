"""GPU wall time of whole solves with and without the two launch-side shortcuts: grid_solve's loop evaluated on the
device (mgx_level_solve) and gs_V recorded into a HIP graph.  Inputs: the reference's test6.input (bottom-up, config 1)
from the golden fixture, the same with `method v-cycle`, and a 128^3 config-3 analogue (v-cycle)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import config3_input            # noqa: E402
from multigrid_amd import cycle, mgio      # noqa: E402
from tests.helpers import load             # noqa: E402

t6 = str(load("test6_f64")["input"])
cases = [("test6 bottom-up f32", t6, np.float32), ("test6 bottom-up f64", t6, np.float64),
         ("test6 v-cycle sor 1.7 f64", t6.replace("sor 1.8", "sor 1.7") + "\nmethod v-cycle\n", np.float64),
         ("config 3 at 128^3 v-cycle tol 1e-6 f64", config3_input((128, 128, 128), tol="1.e-6"), np.float64)]
for name, text, dt in cases:
    for fast in (False, True, True):
        IO = mgio.MgIO(text=text)
        top = mgio.iomg_read(IO, dtype=dt)
        s = cycle.Solver(top, print_sums=False, device_loop=fast, use_graphs=fast)
        s.ctx.sync()
        t0 = time.perf_counter()
        try:
            s.mg_gs_cds(mgio.read_method(IO))
            status = "ok"
        except Exception as e:
            status = f"stopped: {str(e)[:60]}"
        s.ctx.sync()
        dtm = time.perf_counter() - t0
        print(f"{name}: device loop + graph {'on ' if fast else 'off'}  {dtm * 1e3:9.1f} ms  sweeps {[g.itt for g in top.levels()]} {status}", flush=True)
        s.close()
