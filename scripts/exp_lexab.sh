#!/bin/bash
# On the GPU box: the reference-order kernel of the regular build against experiment builds multigrid_amd/libmgx_<name>.so
# (scripts/build_variant.sh <name> "-D..." mgx) named in LIBS: bit-for-bit check against the launch-per-hyperplane form,
# and time per sweep
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for name in "" ${LIBS:-}; do
  echo "== ${name:-regular build}"
  lib=${name:+$R/multigrid_amd/libmgx_$name.so}
  for a in "64 40 f64" "100 40 f64 odd" "256 40 f64" "256 40 f32" "131 40 f32 odd" "512 40 f64" "512 40 f32"; do
    MGX_LIB=$lib timeout -k 10 200 python3 $R/scripts/lex_chain_check.py $a || exit 1
  done
done
