"""Instruction histogram of a line range of a hipcc -S listing (loop bodies): python tools/isa_hist.py file.s first last [top]"""
import collections
import sys

L = open(sys.argv[1]).read().split("\n")
a, b = int(sys.argv[2]) - 1, int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
cnt = collections.Counter()
for ln in L[a:b]:
    ln = ln.strip()
    if not ln or ln[0] in ";." or ln.endswith(":"):
        continue
    cnt[ln.split()[0]] += 1
cls = collections.Counter()
for op, n in cnt.items():
    k = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else "vmem"
    cls[k] += n
print(sum(cnt.values()), dict(cls))
for op, n in cnt.most_common(top):
    print(f"{n:5d} {op}")
