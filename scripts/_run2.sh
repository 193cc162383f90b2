export PYTHONPATH=$PWD
timeout -k 10 300 python -u -m pytest tests/test_gpu_operators.py tests/test_gpu_solve.py -m gpu -x -q --timeout 120 --timeout-method thread -k "not solve_80_v" > gpurun_out/t2.log 2>&1 || { tail -20 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
for lib in multigrid_amd/libmgx.so gpurun_tmp/d3.so; do
export MGX_LIB=$PWD/$lib
for cfg in "20 8" "12 8"; do
set -- $cfg
echo "== $lib TY=$1 NW=$2" >> gpurun_out/q2.log
for b in none all; do for r in 0 3; do MGX_FUSED_TY=$1 MGX_FUSED_NW=$2 PER_SWEEP=1 timeout -k 10 120 python scripts/quick_sweep_timing.py 512 2 f64 $r $b 2>&1 | tail -2 | head -1 >> gpurun_out/q2.log || exit 1; done; done
done
done
cat gpurun_out/q2.log
