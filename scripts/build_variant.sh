#!/bin/bash
# experiment builds: scripts/build_variant.sh NAME "-DFLAG=.. ..." [unit ...]  ->  multigrid_amd/libmgx_NAME.so
# (the named translation units are compiled with the extra flags, the others are taken from the regular build;
#  select the library at run time with MGX_LIB=multigrid_amd/libmgx_NAME.so)
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
C=$HERE/multigrid_amd/csrc
NAME=$1; FLAGS=$2; shift 2
UNITS=${@:-mgx_fused_k}
O=$C/_build/var_$NAME
mkdir -p $O
OBJS=""
for u in mgx mgx_fused_k mgx_tb_k; do
  case " $UNITS " in
    *" $u "*) /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall \
                -Wno-unused-function $FLAGS -c -o $O/$u.o $C/$u.hip & OBJS="$OBJS $O/$u.o";;
    *) OBJS="$OBJS $C/_build/$u.o";;
  esac
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $HERE/multigrid_amd/libmgx_$NAME.so $OBJS
echo built $HERE/multigrid_amd/libmgx_$NAME.so
