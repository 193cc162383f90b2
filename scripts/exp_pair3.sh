#!/bin/bash
# On the GPU box: pair passes at 512^3 fp64, box-free level only, for a list of library builds
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for lib in $1; do
  export MGX_LIB=$R/multigrid_amd/$lib
  for red in ${REDS:-1 0}; do
    echo -n "$lib ${EXP_ENV:-} reduce=$red: "
    env ${EXP_ENV:-} PAIR=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 0 f64 $red none 2>&1 | grep -E "ms/sweep|rror" | tail -1 || exit 1
  done
done
