# usage (GPU box): bash tools/pmc_sq.sh  -> SQ wave-cycle breakdown of the attention kernels (tools/attn_bench.py, p = 0.2544)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_sq -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py 2 > /tmp/pmc_sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_sq/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "attn" in k and "true" in k:
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    wc = sum(d["SQ_WAVE_CYCLES"]) / len(d["SQ_WAVE_CYCLES"])
    for c, v in d.items():
        m = sum(v) / len(v)
        print(f"   {c:28s} {m:16.0f}  {m / wc:6.3f} of wave cycles")
PY
