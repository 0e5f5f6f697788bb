# round-5 GPU call 5: grouped weight-gradient hand-over (one event pair per block), C5 kernel profile
O=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for i in 1 2 3; do
  echo "$(SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/sim8.txt
done
echo "$(SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/sim8.txt
for hm in 1 0 1 0; do
  V1T_HEADMAX8=$hm python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_hm$hm.json
  python - <<PY
import json
d=json.load(open("$O/bench_c5_hm$hm.json")); print("c5 headmax8=$hm", d["value"], d["ms_per_step"])
PY
done
python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
python bench.py --path module --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_module.json
for f in bench_c2 bench_c2_module; do python - <<PY
import json
d=json.load(open("$O/$f.json")); print("$f", d["value"], d["ms_per_step"], d["config"].get("step_path"))
PY
done
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/sim_scaling.txt
cat $O/sim_scaling.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/ks_c5.log 2>&1
cp $(ls /tmp/ks_c5/*/*kernel_stats.csv | head -1) $O/kernel_stats_c5.csv
python3 $GRAFT_REPO_ROOT/tools/prof_top.py $O/kernel_stats_c5.csv 7 8
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 8 1 > $O/rank8.txt 2>&1
