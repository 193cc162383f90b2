export PYTHONPATH=$PWD
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "not solve_80_v" > gpurun_out/t1.log 2>&1 && \
for b in none all; do for r in 0 1 3; do PER_SWEEP=1 timeout -k 10 120 python scripts/quick_sweep_timing.py 512 2 f64 $r $b >> gpurun_out/q1.log 2>&1 || exit 1; done; done
tail -3 gpurun_out/t1.log; cat gpurun_out/q1.log
