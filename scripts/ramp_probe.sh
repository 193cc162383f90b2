#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ramp
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $R/scripts/ramp_probe.py > $OUT/log 2>&1 || { echo "failed"; tail -5 $OUT/log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_tb_sweep" in r["Kernel_Name"]]
d = d[25:]   # the child level's 25 passes come first
lab = ["warm"] * 6 + ["A"] * 5 + ["B"] * 3 + ["C"] * 3 + ["D"] * 3 + ["E"] * 3
for l, v in zip(lab, d): print(l, "%.1f" % v)
PY
rm -rf $OUT/tr
