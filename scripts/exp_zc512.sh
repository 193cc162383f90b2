#!/bin/bash
# On the GPU box: pair passes at 512^3 fp64 (bench boxes, no reductions) for z chunk lengths and tail fractions
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for cfg in "0 -1" "64 12" "128 12" "128 25" "128 33" "128 50" "86 12" "103 12" "103 30" "171 33" "256 50"; do
  set -- $cfg
  echo -n "zc=$1 tail=$2: "
  MGX_FUSED_ZC=$1 MGX_FUSED_TAIL=$2 PAIR=1 MGX_PAIR_PASS=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 0 f64 0 all 2>&1 | grep -E "ms/sweep|rror" | tail -1 | sed 's/.*: //'
done
