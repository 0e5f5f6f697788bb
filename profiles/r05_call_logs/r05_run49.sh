# round-5 GPU call 49: non-temporal stores of the planes only the backward reads (LayerNorm outputs, gelu', fused-MLP activation) against plain stores
# (libv1t_amd_nont.so = -DV1T_NO_NT_SAVED)
O=$GRAFT_REPO_ROOT/gpurun_out/r05ai
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_nont.so; do
    echo "bench $lib: $(V1T_LIB=$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab_nt.txt
  done
done
for lib in libv1t_amd.so libv1t_amd_nont.so; do
  echo "sim4 $lib: $(V1T_LIB=$lib SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_nt.txt
  echo "sim8 $lib: $(V1T_LIB=$lib SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_nt.txt
  echo "module $lib: $(V1T_LIB=$lib python bench.py --path module --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab_nt.txt
done
echo done
