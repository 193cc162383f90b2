#!/bin/bash
# On the GPU box: one sweep in the reference's order (MGX_KERNEL_LEX) with both reductions, fp64, with and without boxes
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for n in ${1:-64 128 256 512}; do
  for b in all none; do
    PER_SWEEP=1 timeout -k 10 200 python3 $R/scripts/quick_sweep_timing.py $n 4 f64 3 $b | grep "ms/sweep" | tail -1 || exit 1
  done
done
