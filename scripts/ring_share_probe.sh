#!/bin/bash
# On the GPU box: FETCH_SIZE per pair pass for several level shapes (one round of blocks, several rounds)
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for sh_ in ${SHAPES:-512,512,64 512,512,128 512,512,256 512,512,512 128,512,512 512,88,512}; do
  shape=${sh_//,/ }
  rm -rf /tmp/pmc_ring
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_ring -- python3 $R/scripts/ring_share_probe.py $shape > /tmp/pmc_ring.log 2>&1 || { echo "pmc failed"; tail -3 /tmp/pmc_ring.log; exit 1; }
  python3 - $shape <<'PY'
import csv, glob, sys
n = [int(v) for v in sys.argv[1:4]]
tot = k = 0
for f in glob.glob("/tmp/pmc_ring/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tb_sweep" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); k += 1
field = n[0] * n[1] * n[2] * 8
import os
print(f"XCD={os.environ.get('MGX_FUSED_XCD', '8')} {n}: fetched {2 * tot / max(k, 1) * 1024 / field:.3f} x the field per pass ({k} passes)")
PY
done
