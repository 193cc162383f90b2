#!/bin/bash
# collect PMC counters for the fused sweep kernel (run on the GPU box via gpurun); one pass per counter group
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
ARGS="${PMC_ARGS:-512 2 f64 1 none}"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  case " ${PMC_ONLY:-1 2 3 4 5 6} " in *" $i "*) ;; *) continue;; esac
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/scripts/quick_sweep_timing.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "k_fused_sweep" not in kn and "k_tb_sweep" not in kn: continue
        key = (kn[kn.index("k_"):][:34], row["Counter_Name"])
        tot[key][0] += float(row["Counter_Value"]); tot[key][1] += 1
for k in sorted(tot): print(k, "avg per launch %.4g over %d launches" % (tot[k][0] / tot[k][1], tot[k][1]))
PY
