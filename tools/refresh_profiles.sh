# usage (GPU box): bash tools/refresh_profiles.sh  -> gpurun_out/${RND:-r05}_*: everything profiles/ holds for this round (copy over afterwards)
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
bash $GRAFT_REPO_ROOT/tools/pmc_bench.sh > $O/pmc_bench.log 2>&1
cp $O/${RND:-r05}_pmc_attention.json $GRAFT_REPO_ROOT/profiles/${RND:-r05}_pmc_attention.json   # bench.py reads it below
bash $GRAFT_REPO_ROOT/tools/pmc_sq.sh > $O/${RND:-r05}_pmc_attention_sq_cycles.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/pmc_lds.sh > $O/${RND:-r05}_pmc_attention_lds.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > /tmp/prof.log 2>&1
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/${RND:-r05}_bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | tail -1 > $O/${RND:-r05}_bench_c2.json
python bench.py --config c1 2>/dev/null | tail -1 > $O/${RND:-r05}_bench_c1.json
python bench.py --config c4 2>/dev/null | tail -1 > $O/${RND:-r05}_bench_c4.json
python bench.py --config c5 2>/dev/null | tail -1 > $O/${RND:-r05}_bench_c5.json
python bench.py --path module 2>/dev/null | tail -1 > $O/${RND:-r05}_bench_c2_module.json
python bench.py --path module-fused --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RND:-r05}_bench_c2_module_fused.json
python bench.py --config c5 --rollout full --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RND:-r05}_bench_c5_full_chain.json
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/${RND:-r05}_sim_scaling.txt
wc -c $O/${RND:-r05}_*
