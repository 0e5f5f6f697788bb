# usage (GPU box): bash tools/pmc_bench.sh  -> per-launch FETCH_SIZE / WRITE_SIZE (KiB) of the attention kernels inside bench.py
# (112-image launches), separate --pmc passes as MI355X_MICROARCH.md prescribes; FETCH_SIZE needs x2 on gfx950
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcb_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmcb_$C.log 2>&1
  python3 - "$C" <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
f = glob.glob(f"/tmp/pmcb_{c}/*/*counter_collection.csv")
if not f:
    print("no counter file", glob.glob(f"/tmp/pmcb_{c}/*/*")); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r.get("Counter_Name") == c and "attn" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{c:10s} {k:70s} n={len(v):3d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
PY
done
