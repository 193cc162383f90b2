export PYTHONPATH=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cls
PER_SWEEP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cls -- python3 $GRAFT_REPO_ROOT/scripts/quick_sweep_timing.py 512 2 f64 ${RED:-0} all > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/cls/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "k_fused" in r["Kernel_Name"]][-6:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows: print(r["Kernel_Name"][:40], int(r["Grid_Size_X"])//576, (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, r["Scratch_Size"])
PY
