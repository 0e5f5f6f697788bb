# usage (GPU box): bash tools/trace_share.sh <world> <rank> -> gpurun_out/share_<world>_<rank>/{kernel_stats.txt,timeline.txt}: one rank's share of an N-GPU step
W=${1:-8}; R=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/share_${W}_${R}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sh_$W
export SIM_ONLY=$W,$R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sh_$W -- python3 $GRAFT_REPO_ROOT/tools/sim_scaling.py > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_top.py $(ls /tmp/sh_$W/*/*kernel_stats.csv | head -1) 7 50 > $OUT/kernel_stats.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $(ls /tmp/sh_$W/*/*kernel_trace.csv | head -1) 2 8 > $OUT/timeline.txt 2>&1
tail -1 $OUT/run.log
