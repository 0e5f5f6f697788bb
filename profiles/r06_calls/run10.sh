# round-6 GPU call 10: attention forward output plane and dK / dV through non-temporal stores against plain stores (experiment build, V1T_ATTN_NT_OUT=0)
O=$GRAFT_REPO_ROOT/gpurun_out/r06j
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -k "attention or gemm_tn or batched_backward" > $O/pytest.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest.txt | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dq2', d.get('roofline_hbm',{}).get('avg_ms'), 'dkv2', d['roofline']['avg_ms'])"; }
for i in 1 2 3 4; do
  echo "native out-nt   : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
  echo "native out-pln  : $(V1T_LIB=libv1t_amd_exp.so V1T_ATTN_NT_OUT=0 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
done
for i in 1 2; do
  echo "module out-nt   : $(python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
  echo "module out-pln  : $(V1T_LIB=libv1t_amd_exp.so V1T_ATTN_NT_OUT=0 python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
done
