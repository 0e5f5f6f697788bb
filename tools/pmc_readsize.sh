# usage (GPU box): RND=r06 bash tools/pmc_readsize.sh   -> gpurun_out/${RND}_pmc_readsize.json
# Calibration of FETCH_SIZE per kernel (MI355X_MICROARCH.md: "FETCH_SIZE = TCC_EA0_RDREQ x 64 B ... other access widths are uncalibrated:
# calibrate on a known byte count in your own access pattern"): the L2's fabric-side read requests BY SIZE - TCC_EA0_RDREQ_32B / _64B / _128B
# (+ the total) - of every kernel of a bench.py step, two counters per pass. Exact read bytes = 32 n32 + 64 n64 + 128 n128.
export RND=${RND:-r06}
ARGS=${@:-"--steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-pmc"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prs_*
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  V1T_DW_SIDE=0 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prs_$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/prs_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os, re, collections
root, rnd = os.environ["GRAFT_REPO_ROOT"], os.environ["RND"]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).strip()
out = collections.defaultdict(dict)
for i in (1, 2, 3):
    f = glob.glob(f"/tmp/prs_{i}/*/*counter_collection.csv")
    if not f:
        print("no counter file for pass", i, open(f"/tmp/prs_{i}.log").read()[-800:]); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = sum(v) / len(v)
res = {}
for k, e in out.items():
    n, n32, n64, n128 = (e.get(f"TCC_EA0_RDREQ{t}_sum", 0.0) for t in ("", "_32B", "_64B", "_128B"))
    w, w64 = e.get("TCC_EA0_WRREQ_sum", 0.0), e.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    res[k] = {"rdreq": n, "rdreq_32b": n32, "rdreq_64b": n64, "rdreq_128b": n128, "read_bytes_by_size": 32 * n32 + 64 * n64 + 128 * n128,
              "fetch_size_equiv_bytes": 64 * n, "wrreq": w, "wrreq_64b": w64, "write_bytes_by_size": 64 * w64 + 32 * (w - w64)}
json.dump({"source": "tools/pmc_readsize.sh: TCC_EA0_RDREQ by request size, mean per launch, V1T_DW_SIDE=0", "kernels": res},
          open(os.path.join(root, "gpurun_out", f"{rnd}_pmc_readsize.json"), "w"), indent=1)
for k, e in sorted(res.items(), key=lambda kv: -kv[1]["read_bytes_by_size"])[:28]:
    print(f"{k[:58]:58s} rd {e['rdreq']:11.0f} = 32B {e['rdreq_32b']:10.0f} + 64B {e['rdreq_64b']:10.0f} + 128B {e['rdreq_128b']:10.0f} -> {e['read_bytes_by_size']/2**20:8.1f} MiB (x64: {e['fetch_size_equiv_bytes']/2**20:8.1f}) wr {e['write_bytes_by_size']/2**20:8.1f} MiB")
PY
