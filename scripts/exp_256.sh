#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for cfg in $1; do
  envs=$(echo $cfg | tr ',' ' ')
  echo -n "$cfg: "
  env $envs PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py ${N:-256} 2 f64 1 all | grep "ms/sweep" | tail -1 | awk '{print $6}' || exit 1
done
