# round-5 GPU call 1: the full GPU suite, the headline bench, the drop-in (module) path, the per-config kernel stats, one rank's share
O=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
python bench.py --path module --no-cpu-baseline 2>$O/bench_c2_module.err | tail -1 > $O/bench_c2_module.json
python bench.py --path module-fused --no-cpu-baseline 2>$O/bench_c2_module_fused.err | tail -1 > $O/bench_c2_module_fused.json
cut -c1-400 $O/bench_c2.json $O/bench_c2_module.json $O/bench_c2_module_fused.json
cd /tmp && export TMPDIR=/tmp
for cfg in c1 c4 c5; do
  rm -rf /tmp/ks_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/ks_$cfg.log 2>&1
  cp $(ls /tmp/ks_$cfg/*/*kernel_stats.csv | head -1) $O/kernel_stats_$cfg.csv
done
rm -rf /tmp/ks_mod
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_mod -- python3 $GRAFT_REPO_ROOT/bench.py --path module --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/ks_mod.log 2>&1
cp $(ls /tmp/ks_mod/*/*kernel_stats.csv | head -1) $O/kernel_stats_c2_module.csv
f=$(ls /tmp/ks_mod/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_kernels.py $f > $O/module_step_kernels.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $f 2 0 > $O/module_step_timeline.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 8 1 > $O/rank8.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/sim_scaling.txt
cat $O/sim_scaling.txt
