# usage (GPU box): bash tools/gap_trace.sh <tag>  -> kernel timeline gaps of bench.py (which kernels the GPU idles after)
TAG=${1:-gap}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/g_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/g_$TAG.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/gap_report.py /tmp/g_$TAG/*/*kernel_trace.csv
tail -1 /tmp/g_$TAG.log | cut -c1-160
