"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace csv: total idle per (previous kernel -> next kernel)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
# last 4 steps only: cut at the last 60 % of the timeline
t0 = ev[0][0] + int(0.45 * (ev[-1][1] - ev[0][0]))
ev = [e for e in ev if e[0] >= t0]
busy = sum(e[1] - e[0] for e in ev)
span = ev[-1][1] - ev[0][0]
gaps = defaultdict(lambda: [0, 0])
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
end, prev = ev[0][1], ev[0][2]
for s, e, n in ev[1:]:
    if s > end:
        k = (short(prev), short(n))
        gaps[k][0] += s - end
        gaps[k][1] += 1
    if e >= end:
        end, prev = e, n
print(f"span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {100 * (1 - busy / span):.1f} % ({len(ev)} kernels)")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"{t / 1e3:9.1f} us in {c:4d} gaps ({t / c / 1e3:6.1f} us each)  {a}  ->  {b}")
