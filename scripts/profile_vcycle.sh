#!/bin/bash
# On the GPU box: per-kernel time of a whole v-cycle solve (rocprofv3 kernel trace summary) -> gpurun_out/prof_vc/
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_vc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fortran-host --vcycle-tol ${VC_TOL:-1.e-6} > $OUT/stats.log 2>&1 || { echo "rocprof failed"; tail -5 $OUT/stats.log; exit 1; }
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1)
cut -c1-60 --complement $f > /dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>8s} total_ms {float(r['TotalDurationNs'])/1e6:10.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
tail -c 700 $OUT/stats.log
