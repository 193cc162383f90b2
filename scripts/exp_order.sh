#!/bin/bash
# On the GPU box: dispatch order / box cut of the fused sweep's piece list; single sweeps with both reductions, 512^3 fp64
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
CFGS=${1:-0,1 1,1 1,2 1,4 2,1 2,2 2,4 0,2}
for cfg in $CFGS; do
  o=${cfg%%,*}; c=${cfg#*,}
  echo -n "order $o boxcut $c single: "
  MGX_FUSED_ORDER=$o MGX_FUSED_BOXCUT=$c PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 2 f64 3 all | grep "ms/sweep" | tail -1 | awk '{print $6}' || exit 1
  echo -n "order $o boxcut $c pair:   "
  MGX_FUSED_ORDER=$o MGX_FUSED_BOXCUT=$c PAIR=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 0 f64 1 all | grep "ms/sweep" | tail -1 | awk '{print $6}' || exit 1
done
