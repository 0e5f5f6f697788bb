# round-5 GPU call 18: lean inference forward (eval / rollout), suite
O=$GRAFT_REPO_ROOT/gpurun_out/r05q
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for i in 1 2 3; do
  python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json
  python - <<PY
import json
d=json.load(open("$O/bench_c5.json")); print("c5", d["value"], d["ms_per_step"])
PY
done
python tools/eval_bench.py 2>/dev/null | tail -3
python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | cut -c1-200
