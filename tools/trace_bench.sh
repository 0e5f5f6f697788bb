# usage (GPU box): bash tools/trace_bench.sh <tag> -> gpurun_out/trace_<tag>/{kernel_trace.csv,gaps.txt,bench.log}: per-launch start / end timestamps of bench.py
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/t_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > $OUT/bench.log 2>&1
cp $(ls /tmp/t_$TAG/*/*kernel_trace.csv | head -1) $OUT/kernel_trace.csv
python3 $GRAFT_REPO_ROOT/tools/gap_report.py $OUT/kernel_trace.csv > $OUT/gaps.txt 2>&1
tail -1 $OUT/bench.log | cut -c1-300
head -30 $OUT/gaps.txt
