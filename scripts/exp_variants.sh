#!/bin/bash
# On the GPU box: time the experiment builds of the library (scripts/build_variant.sh) on the bench-like 512^3 level.
# usage: scripts/exp_variants.sh "name[:ENV=val,ENV=val] ..." ; results in gpurun_out/exp.log
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
mkdir -p $R/gpurun_out
LOG=$R/gpurun_out/exp.log
: > $LOG
for spec in $1; do
  name=${spec%%:*}; envs=""
  if [ "$spec" != "$name" ]; then envs=$(echo ${spec#*:} | tr ',' ' '); fi
  lib=$R/multigrid_amd/libmgx_$name.so
  [ "$name" = base ] && lib=$R/multigrid_amd/libmgx.so
  for args in ${EXP_ARGS:-"512,2,f64,3,all" "512,2,f64,3,none"}; do
    a=$(echo $args | tr ',' ' ')
    echo "== $spec $a" >> $LOG
    env MGX_LIB=$lib $envs timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py $a 2>&1 | grep -E "ms/sweep|rror" | tail -1 >> $LOG || { echo "FAILED $spec" >> $LOG; exit 1; }
  done
done
cat $LOG
