# round-5 GPU call 17: dQ GEMM with four dS' stages in flight
O=$GRAFT_REPO_ROOT/gpurun_out/r05p
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_longseq.py -m gpu -x -q 2>&1 | tail -3
for B in 112 14; do
  for deep in 1 0 1 0; do
    echo "B=$B deep=$deep: $(ATTN_B=$B V1T_DQ2_DEEP=$deep python tools/attn_bench.py 20 2>/dev/null | grep -i "dq_gemm" | tr '\n' ' ')" | tee -a $O/ab_dq2.txt
  done
done
for i in 1 2 3; do
  for deep in 1 0; do
    V1T_DQ2_DEEP=$deep python bench.py --no-cpu-baseline --no-pmc --profile-class 1 2>/dev/null | tail -1 > $O/b.json
    python - <<PY | tee -a $O/ab_dq2_c2.txt
import json
d=json.load(open("$O/b.json")); print("c2 deep=$deep", d["value"], d["ms_per_step"], "dq2 avg ms", d["roofline"]["avg_ms"])
PY
  done
done
