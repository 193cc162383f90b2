#!/bin/bash
# On the GPU box: the bench line, the rocprofv3 kernel-trace summary of the same command, and the HBM traffic
# counters (one pass each) of the sweep kernel.  Results land in gpurun_out/prof/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
BARGS="--steps 50 --warmup 5 --no-vcycle --no-cpu-baseline"
timeout -k 10 300 python3 $R/bench.py $BARGS > $OUT/bench_short.json 2> $OUT/bench_short.err || { echo "bench failed"; tail -5 $OUT/bench_short.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $BARGS > $OUT/stats.log 2>&1 || { echo "rocprof stats failed"; tail -5 $OUT/stats.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 10 --warmup 2 --no-vcycle --no-cpu-baseline > $OUT/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $OUT/pmc_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = rows[:12]
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_fused_sweep" in row["Kernel_Name"]:
            k = row["Counter_Name"]
            tot[k][0] += float(row["Counter_Value"]); tot[k][1] += 1
# one sweep = the launches of all tile classes; count launches of the no-box class = number of sweeps
n_sweeps = None
for f in glob.glob("$OUT/pmc_FETCH_SIZE/**/*counter_collection.csv", recursive=True):
    n_sweeps = sum(1 for row in csv.DictReader(open(f)) if "k_fused_sweep" in row["Kernel_Name"])
out["pmc_totals"] = {k: v[0] for k, v in tot.items()}
out["pmc_launches"] = {k: v[1] for k, v in tot.items()}
out["n_sweeps_seen"] = n_sweeps
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out["pmc_totals"]), out["pmc_launches"], n_sweeps)
PY
cat $OUT/bench_short.json
