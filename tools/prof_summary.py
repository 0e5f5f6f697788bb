"""Summarise a rocprofv3 kernel_stats.csv: python tools/prof_summary.py <csv> [steps] [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
top = int(sys.argv[3]) if len(sys.argv) > 3 else 26
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / steps / 1e6:.2f} ms/step")
for r in rows[:top]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:8.1f} us {float(r['TotalDurationNs']) / steps / 1e6:6.2f} ms/step")
