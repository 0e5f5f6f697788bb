# round-6 GPU call 19: dK/dV kernel reading its own K / V rows non-temporally against plain loads (experiment build, V1T_DKV_KV_NT=0)
O=$GRAFT_REPO_ROOT/gpurun_out/r06r
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_longseq.py -m gpu -q -k "attention or longseq or long" > $O/pytest.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest.txt | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dq2', d.get('roofline_hbm',{}).get('avg_ms'), 'dkv2', d['roofline']['avg_ms'])"; }
for i in 1 2 3 4; do
  echo "kv-nt   : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "kv-pln  : $(V1T_LIB=libv1t_amd_exp.so V1T_DKV_KV_NT=0 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
done
for i in 1 2; do
  echo "module kv-nt   : $(python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "module kv-pln  : $(V1T_LIB=libv1t_amd_exp.so V1T_DKV_KV_NT=0 python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
done
