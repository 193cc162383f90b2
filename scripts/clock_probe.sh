#!/bin/bash
# On the GPU box: does the time of a pass depend on the values through the clock?  Busy cycles (GRBM_GUI_ACTIVE, summed
# over the 8 XCDs) and the wall time per launch of the pair pass and the single sweep on two states: noise and set-up.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/clock
rm -rf $OUT && mkdir -p $OUT
for k in 3 2; do for st in none all; do
  timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 $k f64 1 $st > $OUT/t_${k}_$st.log 2>&1 || { echo "timing failed"; exit 1; }
  tail -2 $OUT/t_${k}_$st.log | head -1
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p_${k}_$st -- python3 $R/scripts/quick_sweep_timing.py 512 $k f64 1 $st > $OUT/p_${k}_$st.log 2>&1 || { echo "pmc failed"; tail -3 $OUT/p_${k}_$st.log; exit 1; }
done; done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p_*_*")):
    if not d.endswith(("none", "all")): continue
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "k_fused_sweep" in kn or "k_tb_sweep" in kn:
                t = tot[kn[kn.index("k_"):][:30]]; t[0] += float(row["Counter_Value"]); t[1] += 1
    for k, v in tot.items(): print(d.split("/")[-1], k, "GRBM_GUI_ACTIVE per launch %.4g over %d" % (v[0] / v[1], v[1]))
PY
