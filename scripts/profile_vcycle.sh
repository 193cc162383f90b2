#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace summary of a whole v-cycle solve of the bench input (tolerance $1, default 1.e-6).
set -u
R=$GRAFT_REPO_ROOT
TOL=${1:-1.e-6}
OUT=$R/gpurun_out/vprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fortran-host --vcycle-tol $TOL > $OUT/stats.log 2>&1 || { echo "rocprof failed"; tail -5 $OUT/stats.log; exit 1; }
f=$(ls $OUT/stats/*/*kernel_stats.csv | tail -1)
cp $f $OUT/vcycle_kernel_stats.csv
# the same trace by kernel AND grid size (a sweep kernel runs on several levels of the hierarchy)
t=$(ls $OUT/stats/*/*kernel_trace.csv | tail -1)
python3 - "$t" "$OUT/vcycle_kernel_by_grid.csv" <<PY
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    a = acc[k]
    a[0] += 1
    a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size_X", "Workgroup_Size_X", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for (n, g, wg), (c, d) in rows:
        w.writerow([n, g, wg, c, int(d), round(d / c, 1), round(100 * d / tot, 3)])
for (n, g, wg), (c, d) in rows[:12]:
    print(n[:48].ljust(48), str(g).rjust(9), str(c).rjust(7), f"{d / c / 1e3:9.1f} us", f"{100 * d / tot:6.2f}%")
PY
rm -rf $OUT/stats
cut -c1-200 $OUT/vcycle_kernel_stats.csv | head -30
tail -1 $OUT/stats.log | cut -c1-1500
