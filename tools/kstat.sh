# usage (GPU box): bash tools/kstat.sh [pattern]  -> per-kernel ms/step of bench.py (5 steps), filtered by pattern
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-pmc > /tmp/b.log 2>&1
f=$(ls /tmp/prof/*/*kernel_stats.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/prof_top.py $f 7 60 | grep -E "${1:-.}"
