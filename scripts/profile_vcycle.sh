#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace summary of a whole v-cycle solve of the bench input (tolerance $1, default 1.e-6).
set -u
R=$GRAFT_REPO_ROOT
TOL=${1:-1.e-6}
OUT=$R/gpurun_out/vprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fortran-host --vcycle-tol $TOL > $OUT/stats.log 2>&1 || { echo "rocprof failed"; tail -5 $OUT/stats.log; exit 1; }
f=$(ls $OUT/stats/*/*kernel_stats.csv | tail -1)
cp $f $OUT/vcycle_kernel_stats.csv
rm -rf $OUT/stats
cut -c1-200 $OUT/vcycle_kernel_stats.csv | head -30
tail -1 $OUT/stats.log | cut -c1-1500
