#!/bin/bash
# On the GPU box: pair passes (two sweeps per pass) at 512^3 fp64 for a list of library builds: with and without the
# bench's boxes, with the error reduction, without reductions and with both
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for lib in $1; do
  export MGX_LIB=$R/multigrid_amd/$lib
  for b in all none; do
    for red in ${REDS:-1 0 3}; do
      echo -n "$lib ${EXP_ENV:-} boxes=$b reduce=$red: "
      env ${EXP_ENV:-} PAIR=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 0 f64 $red $b 2>&1 | grep -E "ms/sweep|rror" | tail -1 || exit 1
    done
  done
done
