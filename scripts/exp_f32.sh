#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 2 f32 3 all | grep "ms/sweep" | tail -1 || exit 1
timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 2 f32 0 all | grep "ms/sweep" | tail -1 || exit 1
PAIR=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 512 3 f32 1 all | grep "ms/sweep" | tail -1 || exit 1
PAIR=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 256 3 f64 1 all | grep "ms/sweep" | tail -1 || exit 1
PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 256 2 f64 1 all | grep "ms/sweep" | tail -1 || exit 1
PER_SWEEP=1 timeout -k 10 120 python3 $R/scripts/quick_sweep_timing.py 128 2 f64 1 all | grep "ms/sweep" | tail -1 || exit 1
