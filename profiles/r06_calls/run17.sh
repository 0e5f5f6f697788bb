# round-6 GPU call 17: dK/dV kernel's tail split (the units of an at most half-filled last round cut 2-4 ways by query range, partial dK / dV summed by a small
# kernel) against whole units only (experiment build, V1T_DKV_SPLIT=0): tests at B = 16 / 6 / 5 / 56, the kernel alone over B, drop-in loop, 2-rank share, native step
O=$GRAFT_REPO_ROOT/gpurun_out/r06p
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -k "attention or golden or native_step_equals or batched" > $O/pytest.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest.txt | tail -4
python tools/dkv2_rounds_probe.py 2>&1 | grep "^B" | tee $O/rounds_split.txt
echo "--- whole units only"; V1T_LIB=libv1t_amd_exp.so V1T_DKV_SPLIT=0 python tools/dkv2_rounds_probe.py 2>&1 | grep "^B" | tee $O/rounds_whole.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dkv2', d['roofline']['avg_ms'])"; }
for i in 1 2 3; do
  echo "module split : $(python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "module whole : $(V1T_LIB=libv1t_amd_exp.so V1T_DKV_SPLIT=0 python bench.py --path module --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
done
for i in 1 2; do
  echo "sim2 split : $(SIM_ONLY=2,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab.txt
  echo "sim2 whole : $(V1T_LIB=libv1t_amd_exp.so V1T_DKV_SPLIT=0 SIM_ONLY=2,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab.txt
  echo "native split : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "native whole : $(V1T_LIB=libv1t_amd_exp.so V1T_DKV_SPLIT=0 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
done
