#!/bin/bash
# On the GPU box: single sweeps with both reductions at 512^3 fp64 under a list of environment settings ("A=1,B=2 A=3 ...")
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for cfg in $1; do
  envs=$(echo $cfg | tr ',' ' ')
  echo -n "$cfg: "
  env $envs PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 2 f64 3 all | grep "ms/sweep" | tail -1 | awk '{print $6}' || exit 1
done
