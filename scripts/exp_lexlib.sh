#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for lib in $1; do
  export MGX_LIB=$R/multigrid_amd/$lib
  for n in 64 256 512; do
    echo -n "$lib n=$n: "
    PER_SWEEP=1 timeout -k 10 200 python3 $R/scripts/quick_sweep_timing.py $n 4 f64 3 all | grep "ms/sweep" | tail -1 | awk '{print $6}' || exit 1
  done
done
