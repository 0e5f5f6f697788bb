# round-5 GPU call 52: non-temporal stores of the QKV plane (ln_gemm_kernel) against plain stores (libv1t_amd_ntqkv.so = -DV1T_NT_QKV)
O=$GRAFT_REPO_ROOT/gpurun_out/r05aj
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_ntqkv.so; do
    echo "bench $lib: $(V1T_LIB=$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab_ntqkv.txt
  done
done
for lib in libv1t_amd.so libv1t_amd_ntqkv.so; do
  echo "sim8 $lib: $(V1T_LIB=$lib SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_ntqkv.txt
  echo "c5 $lib: $(V1T_LIB=$lib python bench.py --config c5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab_ntqkv.txt
done
echo done
