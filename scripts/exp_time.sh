#!/bin/bash
# On the GPU box: single sweeps with both reductions per sweep, and pair passes with the error reduction, 512^3 fp64
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for lib in ${1:-libmgx.so}; do
  export MGX_LIB=$R/multigrid_amd/$lib
  echo "== $lib"
  PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 2 f64 3 all | grep "ms/sweep" | tail -1 || exit 1
  PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 2 f64 3 none | grep "ms/sweep" | tail -1 || exit 1
  PAIR=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 0 f64 1 all | grep "ms/sweep" | tail -1 || exit 1
  PAIR=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 0 f64 1 none | grep "ms/sweep" | tail -1 || exit 1
done
