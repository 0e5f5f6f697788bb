# round-5 GPU call 16: suite on the current build, then the full profile refresh of the round
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_final_pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r05_final_pytest.log
tail -3 gpurun_out/r05_final_pytest.log
RND=r05 bash tools/refresh_profiles.sh > gpurun_out/r05_refresh.log 2>&1
tail -40 gpurun_out/r05_refresh.log
for f in bench_c2 bench_c1 bench_c4 bench_c5 bench_c2_module bench_c2_module_fused; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_$f.json")); print("$f", d["value"], d["ms_per_step"], d["config"].get("step_path"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("traffic_stale"))
except Exception as e: print("$f", "ERR", e)
PY
done
cat gpurun_out/r05_sim_scaling.txt
