#!/bin/bash
# On the GPU box: pair passes at 512^3 fp64 (bench boxes) for a list of library builds: time per sweep, then the bytes
# written and read per launch (WRITE_SIZE, FETCH_SIZE: one counter pass each)
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for lib in $1; do
  export MGX_LIB=$R/multigrid_amd/$lib
  for red in 1 0; do
    echo -n "$lib reduce=$red: "
    PAIR=1 MGX_PAIR_PASS=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 0 f64 $red all 2>&1 | grep -E "ms/sweep|rror" | tail -1 || exit 1
  done
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pmc_$c
    PAIR=1 MGX_PAIR_PASS=1 timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/scripts/quick_sweep_timing.py 512 0 f64 ${PMC_RED:-1} all > /tmp/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -3 /tmp/pmc_$c.log; exit 1; }
    python3 - $c <<'PY'
import csv, glob, sys
c = sys.argv[1]
tot = n = 0
for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tb_sweep" in r["Kernel_Name"] and r["Counter_Name"] == c:
            tot += float(r["Counter_Value"]); n += 1
print(f"   {c}: {tot / max(n, 1) * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e9:.4f} GB per launch ({n} launches)")
PY
  done
done
