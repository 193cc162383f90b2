"""print a rocprofv3 kernel_stats.csv compactly: name (cut), calls, average us, total ms, percent"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6:.3f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(7), f"{float(r['AverageNs']) / 1e3:10.1f} us avg", f"{float(r['TotalDurationNs']) / 1e6:9.3f} ms",
          f"{100 * float(r['TotalDurationNs']) / tot:6.2f}%", f"min {float(r['MinNs']) / 1e3:.1f} max {float(r['MaxNs']) / 1e3:.1f}")
