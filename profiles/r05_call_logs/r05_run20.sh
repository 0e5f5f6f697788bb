# round-5 GPU call 20: weight-gradient GEMM staging by TileDma (A/B against the per-piece form: separate builds)
O=$GRAFT_REPO_ROOT/gpurun_out/r05s
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  for lib in libv1t_amd.so libv1t_amd_notd.so; do
    echo "N=8 $lib: $(V1T_LIB=$lib SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_td.txt
    echo "N=4 $lib: $(V1T_LIB=$lib SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_td.txt
  done
done
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_notd.so; do
    V1T_LIB=$lib python bench.py --no-cpu-baseline --no-pmc --profile-class 4 2>/dev/null | tail -1 > $O/b.json
    python - <<PY | tee -a $O/ab_td.txt
import json
d=json.load(open("$O/b.json")); print("c2 $lib", d["value"], d["ms_per_step"], "gemm_tn avg ms", d["roofline"]["avg_ms"], d["roofline"]["launches"])
PY
  done
done
for lib in libv1t_amd.so libv1t_amd_notd.so; do
  V1T_LIB=$lib python bench.py --path module --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/b.json
  python - <<PY | tee -a $O/ab_td.txt
import json
d=json.load(open("$O/b.json")); print("module $lib", d["value"], d["ms_per_step"])
PY
done
