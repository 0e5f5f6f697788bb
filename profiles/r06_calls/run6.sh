# round-6 GPU call 6: a block's weight-gradient GEMMs as ONE fused launch (gemm_tn2_group_kernel) against one launch per GEMM (experiment build,
# V1T_TN_GROUP_FUSE=0), and the m-chunk target under fusion (V1T_TN_WGS = workgroups per GEMM): tests, native step, drop-in loop, 8- and 4-rank shares
O=$GRAFT_REPO_ROOT/gpurun_out/r06f
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -k "gemm_tn or batched_backward or fused_training or backward_fusions or native_step_equals or golden" > $O/pytest_tn.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest_tn.txt | tail -4
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
E="V1T_LIB=libv1t_amd_exp.so"
for i in 1 2; do
  echo "native fused        : $(python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "native unfused      : $(env $E V1T_TN_GROUP_FUSE=0 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "native fused wgs256 : $(env $E V1T_TN_WGS=256 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "native fused wgs128 : $(env $E V1T_TN_WGS=128 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab.txt
done
for i in 1 2; do
  echo "module fused        : $(python bench.py --path module --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "module unfused      : $(env $E V1T_TN_GROUP_FUSE=0 python bench.py --path module --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab.txt
  for w in 32 64 128; do
    echo "module fused wgs$w  : $(env $E V1T_TN_WGS=$w python bench.py --path module --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab.txt
  done
done
for w in 0 32 64 128; do
  echo "sim8 fused wgs$w : $(env $E V1T_TN_WGS=$w SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab.txt
  echo "sim4 fused wgs$w : $(env $E V1T_TN_WGS=$w SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab.txt
done
echo "sim8 unfused : $(env $E V1T_TN_GROUP_FUSE=0 SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab.txt
echo "sim4 unfused : $(env $E V1T_TN_GROUP_FUSE=0 SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab.txt
