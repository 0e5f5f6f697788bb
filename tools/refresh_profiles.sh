# usage (GPU box): bash tools/refresh_profiles.sh  -> gpurun_out/${RND:-r06}_*: everything profiles/ holds for this round (copy over afterwards)
export RND=${RND:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
bash $GRAFT_REPO_ROOT/tools/pmc_bench.sh > $O/pmc_bench.log 2>&1
cp $O/${RND}_pmc_attention.json $GRAFT_REPO_ROOT/profiles/${RND}_pmc_attention.json   # bench.py reads it below
bash $GRAFT_REPO_ROOT/tools/pmc_sq.sh > $O/${RND}_pmc_attention_sq_cycles.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/pmc_lds.sh > $O/${RND}_pmc_attention_lds.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# kernel-level profiles of every BASELINE config and of the drop-in (module) path: rocprofv3 --kernel-trace --stats of bench.py
for cfg in c2 c1 c4 c5; do
  rm -rf /tmp/prof_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > /tmp/prof_$cfg.log 2>&1
  cp $(ls /tmp/prof_$cfg/*/*kernel_stats.csv | head -1) $O/${RND}_kernel_stats_$cfg.csv
done
cp $O/${RND}_kernel_stats_c2.csv $O/${RND}_bench_kernel_stats.csv
f=$(ls /tmp/prof_c2/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_kernels.py $f > $O/${RND}_step_kernels.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $f 2 25 > $O/${RND}_step_timeline.txt 2>&1
rm -rf /tmp/prof_mod
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mod -- python3 $GRAFT_REPO_ROOT/bench.py --path module --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > /tmp/prof_mod.log 2>&1
cp $(ls /tmp/prof_mod/*/*kernel_stats.csv | head -1) $O/${RND}_kernel_stats_c2_module.csv
f=$(ls /tmp/prof_mod/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_kernels.py $f > $O/${RND}_module_step_kernels.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 8 1 > $O/${RND}_rank8_census.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 4 1 > $O/${RND}_rank4_census.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | tail -1 > $O/${RND}_bench_c2.json
python bench.py --config c1 2>/dev/null | tail -1 > $O/${RND}_bench_c1.json
python bench.py --config c4 2>/dev/null | tail -1 > $O/${RND}_bench_c4.json
python bench.py --config c5 2>/dev/null | tail -1 > $O/${RND}_bench_c5.json
python bench.py --path module 2>/dev/null | tail -1 > $O/${RND}_bench_c2_module.json
python bench.py --path module-fused --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RND}_bench_c2_module_fused.json
python bench.py --config c5 --rollout full --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RND}_bench_c5_full_chain.json
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/${RND}_sim_scaling.txt
python tools/sim_scaling.py 2>/dev/null | grep "^world" >> $O/${RND}_sim_scaling.txt
wc -c $O/${RND}_*
