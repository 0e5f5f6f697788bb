# usage (GPU box): bash tools/pmc_headmax.sh -> SQ wave-cycle breakdown of the rollout head-max kernels (tools/rollout_bench.py, batch 256)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_hm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_hm -- python3 $GRAFT_REPO_ROOT/tools/rollout_bench.py 256 > /tmp/pmc_hm.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_hm/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "headmax" in k or "vecmat" in k:
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    vals = sorted(d["SQ_WAVE_CYCLES"])
    big = vals[len(vals) // 2]  # the dense launches dominate the upper half
    for c, v in d.items():
        v2 = [x for x, w in zip(v, d["SQ_WAVE_CYCLES"]) if w >= big]
        m = sum(v2) / max(len(v2), 1)
        wc = sum(w for w in d["SQ_WAVE_CYCLES"] if w >= big) / max(len(v2), 1)
        print(f"   {c:28s} {m:16.0f}  {m / wc:6.3f} of wave cycles")
PY
