# usage (on the GPU box, via gpurun): bash tools/prof_bench.sh <tag>  -> gpurun_out/prof_<tag>/kernel_stats.csv
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > $OUT/bench.log 2>&1
cp /tmp/p_$TAG/*/*kernel_stats.csv $OUT/kernel_stats.csv
head -40 $OUT/kernel_stats.csv | cut -d, -f1-4
tail -1 $OUT/bench.log
