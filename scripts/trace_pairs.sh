#!/bin/bash
# On the GPU box: kernel trace of a short v-cycle solve; duration of every two-sweep pass grouped by the kernel before it.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ptrace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fortran-host --vcycle-tol ${1:-1.e-4} > $OUT/log 2>&1 || { echo "rocprof failed"; tail -5 $OUT/log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "")
    return n[:n.index("(")] if "(" in n else n
by = collections.defaultdict(list)
seq = []
for i, r in enumerate(rows):
    n = short(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seq.append((n, d, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for i, (n, d, s, e) in enumerate(seq):
    if n.startswith("k_tb_sweep"):
        prev = seq[i - 1][0] if i else "-"
        gap = (s - seq[i - 1][3]) / 1e3 if i else 0
        by[prev].append((d, gap))
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    ds = [a for a, _ in v]; gs = [b for _, b in v]
    print("%-45s n=%5d  avg %.1f us  min %.1f  max %.1f   gap before avg %.1f us" % (k[:45], len(v), sum(ds) / len(ds), min(ds), max(ds), sum(gs) / len(gs)))
# a typical cycle: print 60 consecutive launches from the middle
m = len(seq) // 2
for n, d, s, e in seq[m:m + 70]:
    print("%-50s %8.1f us" % (n[:50], d))
PY
rm -rf $OUT/tr
