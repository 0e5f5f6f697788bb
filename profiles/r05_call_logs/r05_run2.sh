# round-5 GPU call 2: suite + A/Bs of the ring K loop / low-priority dW stream at a rank's share, module path after the sync removal, C5
O=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2; do
  for ring in 1 0; do
    echo "ring=$ring: $(V1T_GEMM_RING=$ring SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_ring.txt
    echo "ring=$ring: $(V1T_GEMM_RING=$ring SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_ring.txt
  done
  for prio in low normal; do
    echo "dwprio=$prio: $(V1T_DW_PRIO=$prio SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_prio.txt
  done
done
python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
python bench.py --path module --no-cpu-baseline 2>$O/bench_c2_module.err | tail -1 > $O/bench_c2_module.json
V1T_GEMM_RING=0 python bench.py --path module --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_module_noring.json
python bench.py --path module-fused --no-cpu-baseline 2>$O/bench_c2_module_fused.err | tail -1 > $O/bench_c2_module_fused.json
python bench.py --config c5 --no-cpu-baseline 2>$O/bench_c5.err | tail -1 > $O/bench_c5.json
for f in bench_c2 bench_c2_module bench_c2_module_noring bench_c2_module_fused bench_c5; do python - <<PY
import json
d=json.load(open("$O/$f.json")); print("$f", d["value"], d["ms_per_step"], d["config"].get("step_path"))
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_mod /tmp/ks_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_mod -- python3 $GRAFT_REPO_ROOT/bench.py --path module --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/ks_mod.log 2>&1
cp $(ls /tmp/ks_mod/*/*kernel_stats.csv | head -1) $O/kernel_stats_c2_module.csv
f=$(ls /tmp/ks_mod/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_kernels.py $f > $O/module_step_kernels.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $f 2 0 > $O/module_step_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/ks_c5.log 2>&1
cp $(ls /tmp/ks_c5/*/*kernel_stats.csv | head -1) $O/kernel_stats_c5.csv
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 8 1 > $O/rank8.txt 2>&1
