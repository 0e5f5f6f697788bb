# round-6 GPU call 15: dS' blocks key-block-major ([bh][key block][query block]: contiguous per-wave store streams in the dK/dV kernel, one contiguous 32 KB per
# workgroup and stage in the dQ GEMM) against the query-block-major layout of rounds 2-5 (experiment build, V1T_DS_KMAJOR=0)
O=$GRAFT_REPO_ROOT/gpurun_out/r06n
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_longseq.py -m gpu -q -k "attention or longseq or long" > $O/pytest.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest.txt | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dq2', d.get('roofline_hbm',{}).get('avg_ms'), 'dkv2', d['roofline']['avg_ms'])"; }
for i in 1 2 3 4; do
  echo "k-major : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "q-major : $(V1T_LIB=libv1t_amd_exp.so V1T_DS_KMAJOR=0 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
done
for i in 1 2; do
  echo "module k-major : $(python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "module q-major : $(V1T_LIB=libv1t_amd_exp.so V1T_DS_KMAJOR=0 python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
done
