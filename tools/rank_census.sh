# usage (GPU box): bash tools/rank_census.sh WORLD RANK  -> launch census of one simulated rank's step (tools/sim_scaling.py SIM_ONLY + tools/step_kernels.py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rank
SIM_ONLY=$1,$2 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rank -- python3 $GRAFT_REPO_ROOT/tools/sim_scaling.py > /tmp/rank.log 2>&1
tail -1 /tmp/rank.log
f=$(ls /tmp/prof_rank/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_kernels.py $f
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $f 2 0
