export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export PER_SWEEP=1
for args in "3 all" "1 all" "0 all" "3 none" "1 none"; do
  timeout -k 10 60 python scripts/quick_sweep_timing.py 512 2 f64 $args 2>&1 | grep "ms/sweep" | tail -1 >> gpurun_out/t.log || exit 1
done
cat gpurun_out/t.log
