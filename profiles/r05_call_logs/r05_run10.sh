# round-5 GPU call 10: grouped slab reductions; head-max SQ counters
O=$GRAFT_REPO_ROOT/gpurun_out/r05j
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/sim_scaling.txt
cat $O/sim_scaling.txt
python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
python bench.py --path module --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_module.json
python bench.py --path module-fused --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_module_fused.json
for f in bench_c2 bench_c2_module bench_c2_module_fused; do python - <<PY
import json
d=json.load(open("$O/$f.json")); print("$f", d["value"], d["ms_per_step"], d["config"].get("step_path"))
PY
done
bash $GRAFT_REPO_ROOT/tools/pmc_headmax.sh > $O/pmc_headmax.txt 2>&1
cat $O/pmc_headmax.txt | tail -24
cd $GRAFT_REPO_ROOT
python tools/sim_scaling.py 2>/dev/null | grep "^world" >> $O/sim_scaling.txt
tail -4 $O/sim_scaling.txt
