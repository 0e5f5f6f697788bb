# round-5 GPU call 6: forward-attention LPT order A/B (separate builds), head-max store ablation
O=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for B in 14 28 112; do
  for lib in libv1t_amd.so libv1t_amd_nolpt.so libv1t_amd.so libv1t_amd_nolpt.so; do
    echo "B=$B $lib: $(ATTN_B=$B V1T_LIB=$lib python tools/attn_bench.py 20 2>/dev/null | grep -i fwd | tr '\n' ' ')" | tee -a $O/ab_lpt.txt
  done
done
for lib in libv1t_amd.so libv1t_amd_hmns.so libv1t_amd.so libv1t_amd_hmns.so; do
  V1T_LIB=$lib python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/c5.json
  python - <<PY
import json
d=json.load(open("$O/c5.json")); print("c5 $lib", d["value"], d["ms_per_step"])
PY
done | tee -a $O/ab_hm_nostore.txt
for lib in libv1t_amd.so libv1t_amd_nolpt.so libv1t_amd.so libv1t_amd_nolpt.so; do
  echo "$lib: $(V1T_LIB=$lib SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_lpt_sim.txt
done
for lib in libv1t_amd.so libv1t_amd_nolpt.so libv1t_amd.so libv1t_amd_nolpt.so; do
  V1T_LIB=$lib python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/c2.json
  python - <<PY
import json
d=json.load(open("$O/c2.json")); print("c2 $lib", d["value"], d["ms_per_step"])
PY
done | tee -a $O/ab_lpt_c2.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
