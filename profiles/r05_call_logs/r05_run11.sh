# round-5 GPU call 11: head-max with the K staging on group 1 only
O=$GRAFT_REPO_ROOT/gpurun_out/r05k
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "rollout or recorder" 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json
  python - <<PY
import json
d=json.load(open("$O/bench_c5.json")); print("c5", d["value"], d["ms_per_step"])
PY
done
bash $GRAFT_REPO_ROOT/tools/pmc_headmax.sh > $O/pmc_headmax.txt 2>&1
tail -20 $O/pmc_headmax.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > $O/ks_c5.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_top.py $(ls /tmp/ks_c5/*/*kernel_stats.csv | head -1) 7 6
