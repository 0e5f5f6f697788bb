"""One training step of a rocprofv3 --kernel-trace csv of bench.py as a launch census: every kernel name launched between two pack_kernel launches (= one optimizer
step), with counts and summed duration. usage: python tools/step_kernels.py kernel_trace.csv [step index from the end, default 2]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
packs = [i for i, e in enumerate(ev) if "pack_kernel" in e[2]]
step = ev[packs[-back - 1]:packs[-back]]
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]  # noqa: E731
cnt, dur = collections.Counter(), collections.Counter()
for s, e, n in step:
    cnt[short(n)] += 1
    dur[short(n)] += e - s
foreign = [n for n in cnt if n.startswith("at::") or n.startswith("void at::") or n.startswith("__amd_rocclr")]
print(f"one step: {len(step)} kernel launches over {(step[-1][1] - step[0][0]) / 1e6:.3f} ms, {len(cnt)} distinct kernels, "
      f"{sum(cnt[n] for n in foreign)} launches of ATen / runtime kernels {foreign if foreign else ''}")
for n, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{c:4d} x {n:72s} {dur[n] / 1e3:9.1f} us")
