#!/bin/bash
# On the GPU box: single sweeps against pair passes on the smaller levels (fp64, bench boxes, error reduction)
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for n in ${SIZES:-256 384 128}; do
  for mode in "PER_SWEEP=1 MGX_PAIR_PASS=0" "PAIR=1 MGX_PAIR_PASS=1" ${EXTRA:-}; do
    for red in 1 0; do
      echo -n "n=$n $mode reduce=$red: "
      env $mode timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py $n 0 f64 $red all 2>&1 | grep -E "ms/sweep|rror" | tail -1 | sed 's/.*: //'
    done
  done
done
