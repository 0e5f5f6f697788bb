# usage (GPU box): bash tools/prof_env.sh <tag> <grep-pattern>   (env vars pass through) -> per-step kernel times of bench.py
TAG=$1; PAT=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > /tmp/p_$TAG.log 2>&1
python3 - "$TAG" "$PAT" <<'PY'
import csv, glob, sys, re
tag, pat = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/p_{tag}/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print(tag, "total ms/step", round(sum(int(r["TotalDurationNs"]) for r in rows) / 7e6, 3))
for r in rows:
    if re.search(pat, r["Name"]):
        print(f"  {r['Name'][:80]:80s} {r['Calls']:>5s} {int(r['TotalDurationNs'])/7e6:7.3f} ms/step {float(r['AverageNs'])/1e3:8.1f} us")
PY
