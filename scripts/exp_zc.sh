#!/bin/bash
# On the GPU box: pair passes (fp64, bench boxes, error reduction) at the sizes given, for several z chunk lengths
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for n in ${SIZES:-256 384}; do
  for zc in ${ZCS:-0 16 24 32 43 64}; do
    echo -n "n=$n zc=$zc: "
    MGX_FUSED_ZC=$zc PAIR=1 MGX_PAIR_PASS=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py $n 0 f64 1 all 2>&1 | grep -E "ms/sweep|rror" | tail -1 | sed 's/.*: //'
  done
done
