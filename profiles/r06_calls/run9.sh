# round-6 GPU call 9: weight-gradient GEMM reading its once-read operand with the non-temporal policy against plain (experiment build, V1T_TN2_NT=0)
O=$GRAFT_REPO_ROOT/gpurun_out/r06i
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -k "attention or gemm_tn or batched_backward" > $O/pytest.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest.txt | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dq2', d.get('roofline_hbm',{}).get('avg_ms'), 'dkv2', d['roofline']['avg_ms'])"; }
for i in 1 2 3 4; do
  echo "native tn-nt    : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
  echo "native tn-plain : $(V1T_LIB=libv1t_amd_exp.so V1T_TN2_NT=0 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
done
for i in 1 2; do
  echo "module tn-nt    : $(python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
  echo "module tn-plain : $(V1T_LIB=libv1t_amd_exp.so V1T_TN2_NT=0 python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_nt.txt
done
