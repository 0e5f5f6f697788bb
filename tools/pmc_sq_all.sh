# usage (GPU box): RND=r06 bash tools/pmc_sq_all.sh  -> gpurun_out/${RND}_pmc_sq_all.json
# SQ wave-cycle breakdown of EVERY kernel of a bench.py step (north_star: "MFMA utilisation against gfx950 peak"): SQ_WAVE_CYCLES, SQ_BUSY_CYCLES,
# SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_VALU / _LDS / _ANY in one pass (8 SQ slots), V1T_DW_SIDE=0 (every kernel alone).
export RND=${RND:-r06}
ARGS=${@:-"--steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-pmc"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/psq_all
V1T_DW_SIDE=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/psq_all -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/psq_all.log 2>&1
python3 - <<'PY'
import csv, glob, json, os, re, collections
root, rnd = os.environ["GRAFT_REPO_ROOT"], os.environ["RND"]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).strip()
f = glob.glob("/tmp/psq_all/*/*counter_collection.csv")
if not f:
    print(open("/tmp/psq_all.log").read()[-2000:]); raise SystemExit(1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = max(m.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    out[k] = dict(m, launches=len(d.get("SQ_WAVE_CYCLES", [])), mfma_busy_per_wave_cycle=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / wc,
                  wait_any_frac=m.get("SQ_WAIT_ANY", 0.0) / wc, wait_inst_frac=m.get("SQ_WAIT_INST_ANY", 0.0) / wc, active_valu_frac=m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                  waves_per_busy_cycle=wc / max(m.get("SQ_BUSY_CYCLES", 0.0), 1.0))
json.dump({"source": "tools/pmc_sq_all.sh: rocprofv3 --pmc (8 SQ counters, one pass), mean per launch, V1T_DW_SIDE=0", "kernels": out},
          open(os.path.join(root, "gpurun_out", f"{rnd}_pmc_sq_all.json"), "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:24]:
    print(f"{k[:56]:56s} MFMA busy {e['mfma_busy_per_wave_cycle']:5.2f}  wait {e['wait_any_frac']:5.2f}  issue-stall {e['wait_inst_frac']:5.2f}  VALU {e['active_valu_frac']:5.2f}  waves/busy-cycle {e['waves_per_busy_cycle']:6.1f}")
PY
