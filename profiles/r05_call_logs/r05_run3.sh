# round-5 GPU call 3: dy double buffer, TN_WGS sweep at a rank's share and on the module path, pipelined rollout
O=$GRAFT_REPO_ROOT/gpurun_out/r05c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for i in 1 2; do
  for wgs in 512 256 128 64; do
    echo "tn_wgs=$wgs: $(V1T_TN_WGS=$wgs SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_tnwgs.txt
  done
  for ring in 1 0; do
    echo "ring=$ring: $(V1T_GEMM_RING=$ring SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_ring.txt
  done
done
for wgs in 512 256 128; do
  V1T_TN_WGS=$wgs python bench.py --path module --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_module_wgs$wgs.json
done
python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
python bench.py --config c5 --no-cpu-baseline 2>$O/bench_c5.err | tail -1 > $O/bench_c5.json
for f in bench_c2 bench_c2_module_wgs512 bench_c2_module_wgs256 bench_c2_module_wgs128 bench_c5; do python - <<PY
import json
d=json.load(open("$O/$f.json")); print("$f", d["value"], d["ms_per_step"], d["config"].get("step_path"))
PY
done
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/sim_scaling.txt
cat $O/sim_scaling.txt
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 8 1 > $O/rank8.txt 2>&1
