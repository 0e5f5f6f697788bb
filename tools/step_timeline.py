"""One step of a rocprofv3 --kernel-trace csv of bench.py as a timeline: main-stream kernels (>= 25 us) with start offset / duration,
side-stream activity summarised. usage: python tools/step_timeline.py kernel_trace.csv [step index from the end, default 3] [min us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in rows))
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]  # noqa: E731
packs = [i for i, e in enumerate(ev) if "pack_kernel" in e[2]]
step = ev[packs[-back - 1]:packs[-back]]
base = step[0][0]
main = max(set(e[3] for e in step), key=lambda q: sum(e[1] - e[0] for e in step if e[3] == q))
print(f"step span {(step[-1][1] - base) / 1e6:.3f} ms, {len(step)} kernels, main stream {main}")
prev_end = base
for s, e, n, q in step:
    if q != main:
        continue
    gap = (s - prev_end) / 1e3
    if gap > 20:
        side = [x for x in step if x[3] != main and x[0] < s and x[1] > prev_end]
        print(f"{'':9s} ---- main idle {gap:7.1f} us; side streams ran {len(side)} kernels ({sum(x[1] - x[0] for x in side) / 1e3:.0f} us of kernel time)")
    if (e - s) / 1e3 >= min_us:
        conc = sorted({short(x[2]) for x in step if x[3] != main and x[0] < e and x[1] > s})
        print(f"{(s - base) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {short(n):44s} {'|| ' + ','.join(conc)[:90] if conc else ''}")
    prev_end = max(prev_end, e)
