# usage (GPU box): bash tools/prof_any.sh <tag> <script.py> [args]  -> prints top kernels by total time
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/p_$TAG.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG && cp /tmp/p_$TAG/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/kernel_stats.csv
head -16 /tmp/p_$TAG/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
tail -2 /tmp/p_$TAG.log
