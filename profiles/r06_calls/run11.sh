# round-6 GPU call 11: A operand of gemm_nt / gemm_lnbwd (dX GEMMs, proj, dO: read once per launch, 64 B of a row per K tile) through non-temporal
# loads (experiment build, V1T_GEMM_A_NT=1) against the plain loads of the product
O=$GRAFT_REPO_ROOT/gpurun_out/r06k
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3 4; do
  echo "native plain : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
  echo "native a-nt  : $(V1T_LIB=libv1t_amd_exp.so V1T_GEMM_A_NT=1 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab.txt
done
