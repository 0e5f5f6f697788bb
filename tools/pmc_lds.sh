# usage (GPU box): bash tools/pmc_lds.sh  -> LDS pipe counters of the attention kernels (tools/attn_bench.py, p = 0.2544), per-kernel means
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_lds -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py 2 > /tmp/pmc_lds.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_lds/*/*counter_collection.csv")
if not f:
    print(open("/tmp/pmc_lds.log").read()[-2000:]); raise SystemExit(1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "attn" in k and ("true" in k or "dq2" in k):
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    base = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"]) if d.get("GRBM_GUI_ACTIVE") else 1.0
    for c, v in d.items():
        m = sum(v) / len(v)
        print(f"   {c:28s} {m:16.0f}  {m / base:10.3f} per GPU-active cycle")
PY
