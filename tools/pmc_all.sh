# usage (GPU box): RND=r06 bash tools/pmc_all.sh [bench args, default: the C2 training step]
# Counter-backed HBM traffic + durations of EVERY kernel of a bench.py step: separate rocprofv3 passes as
# MI355X_MICROARCH.md prescribes - `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (KiB per launch; FETCH_SIZE needs x2 on gfx950: applied by
# tools/roofline_table.py, not here), and two plain `--kernel-trace --stats` passes for the durations: one as the step runs (weight-gradient
# GEMMs beside the main stream's kernels) and one with V1T_DW_SIDE=0 (every kernel alone on the chip: the duration a bandwidth figure needs).
# Output: gpurun_out/${RND}_pmc_all${TAG}.json (copy to profiles/).
export RND=${RND:-r06}
TAG=${TAG:-}
ARGS=${@:-"--steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa_*
for C in FETCH_SIZE WRITE_SIZE; do
  V1T_DW_SIDE=0 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pa_$C -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/pa_$C.log 2>&1
done
V1T_DW_SIDE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_alone -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/pa_alone.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_live -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/pa_live.log 2>&1
python3 - "$ARGS" <<'PY'
import csv, glob, json, os, sys, collections, re
root, rnd, tag = os.environ["GRAFT_REPO_ROOT"], os.environ["RND"], os.environ.get("TAG", "")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).strip()
out = {"bench_args": sys.argv[1], "source": "tools/pmc_all.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB per launch, mean over launches; FETCH_SIZE "
       "still needs x2 on gfx950), durations from two --kernel-trace --stats passes (alone: V1T_DW_SIDE=0, live: the default second stream)", "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pa_{c}/*/*counter_collection.csv")
    if not f:
        print("no counter file for", c, open(f"/tmp/pa_{c}.log").read()[-1500:]); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == c:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        e = out["kernels"].setdefault(k, {})
        e["fetch_kib" if c == "FETCH_SIZE" else "write_kib"] = sum(v) / len(v)
        e["pmc_launches"] = len(v)
        e[("fetch" if c == "FETCH_SIZE" else "write") + "_kib_minmax"] = [min(v), max(v)]
for mode in ("alone", "live"):
    f = glob.glob(f"/tmp/pa_{mode}/*/*kernel_stats.csv")
    if not f:
        print("no stats file for", mode, open(f"/tmp/pa_{mode}.log").read()[-1500:]); continue
    for r in csv.DictReader(open(f[0])):
        e = out["kernels"].setdefault(short(r["Name"]), {})
        e[f"calls_{mode}"] = int(r["Calls"]); e[f"avg_us_{mode}"] = float(r["AverageNs"]) / 1e3; e[f"total_ms_{mode}"] = float(r["TotalDurationNs"]) / 1e6
    os.system(f"cp {f[0]} {root}/gpurun_out/{rnd}_kernel_stats_{mode}{tag}.csv")
    log = open(f"/tmp/pa_{mode}.log").read().strip().splitlines()
    line = next((l for l in reversed(log) if l.startswith("{")), None)
    if line:
        d = json.loads(line); out[f"bench_{mode}"] = {"ms_per_step": d.get("ms_per_step"), "value": d.get("value"), "steps": d.get("steps"), "warmup": d.get("warmup")}
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(root, "gpurun_out", f"{rnd}_pmc_all{tag}.json"), "w"), indent=1)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("total_ms_alone", 0)):
    print(f"{k[:70]:70s} calls {e.get('calls_alone', 0):4d} alone {e.get('avg_us_alone', 0):8.1f} us live {e.get('avg_us_live', 0):8.1f} us fetch {e.get('fetch_kib', 0)/1024:9.1f} MiB write {e.get('write_kib', 0)/1024:9.1f} MiB")
PY
