"""Dev tool: for every __amd_rocclr_* / at::native kernel of a rocprofv3 --kernel-trace csv: stream and the kernels launched before / after it on that stream.
usage: python tools/copy_context.py kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in rows))
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:40]  # noqa: E731
by = collections.defaultdict(list)
for s, n, q in ev:
    by[q].append(n)
cnt = collections.Counter()
for q, names in by.items():
    for i, n in enumerate(names):
        if "rocclr" in n or "at::native" in n:
            prev = short(names[i - 1]) if i else "-"
            nxt = short(names[i + 1]) if i + 1 < len(names) else "-"
            cnt[(short(n), q, prev, nxt)] += 1
for (n, q, p, x), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:5d}  {n:40s} stream {q:>3s}  after {p:40s} before {x}")
