"""Dev tool: top kernels of a rocprofv3 kernel_stats.csv as ms per step. usage: python tools/prof_top.py <csv> [steps=7] [n=40]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", round(tot / steps / 1e6, 3))
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {int(r['TotalDurationNs']) / steps / 1e6:7.3f} ms/step {float(r['AverageNs']) / 1e3:8.1f} us")
