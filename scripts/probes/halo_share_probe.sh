#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for nz in 64 128 512; do
  rm -rf /tmp/hs_$nz
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/hs_$nz -- python3 $R/scripts/probes/halo_share_probe.py $nz > /tmp/hs_$nz.log 2>&1 || { echo "failed $nz"; tail -3 /tmp/hs_$nz.log; exit 1; }
  python3 - <<PY
import csv, glob
tot=0; n=0
for f in glob.glob("/tmp/hs_$nz/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_fused_sweep" in row["Kernel_Name"] and row["Counter_Name"]=="FETCH_SIZE":
            tot+=float(row["Counter_Value"]); n+=1
b=2*tot/n*1024
print("nz=$nz launches",n,"fetch bytes per launch (FETCH_SIZE x2 KiB) %.4g"%b,"ratio to compulsory %.3f"%(b/(512*512*$nz*8)))
PY
done
