# round-5 GPU call 4: 8-wave head-max (C5), reference-Recorder mechanism test, small-launch state after TN_WGS=256
O=$GRAFT_REPO_ROOT/gpurun_out/r05d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for i in 1 2; do
  for hm in 1 0; do
    V1T_HEADMAX8=$hm python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_hm$hm.$i.json
    python - <<PY
import json
d=json.load(open("$O/bench_c5_hm$hm.$i.json")); print("c5 headmax8=$hm", d["value"], d["ms_per_step"])
PY
  done
done
python bench.py --config c5 --rollout full --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
python bench.py --path module --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_module.json
python bench.py --path module-fused --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_module_fused.json
for f in bench_c2 bench_c2_module bench_c2_module_fused; do python - <<PY
import json
d=json.load(open("$O/$f.json")); print("$f", d["value"], d["ms_per_step"], d["config"].get("step_path"))
PY
done
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/sim_scaling.txt
cat $O/sim_scaling.txt
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 8 1 > $O/rank8.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 4 1 > $O/rank4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_mod
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_mod -- python3 $GRAFT_REPO_ROOT/bench.py --path module --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/ks_mod.log 2>&1
f=$(ls /tmp/ks_mod/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_kernels.py $f > $O/module_step_kernels.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $f 2 0 > $O/module_step_timeline.txt 2>&1
