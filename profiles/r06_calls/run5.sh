# round-6 GPU call 5: weight-gradient GEMM with a 3-stage operand ring (two stages in flight) against the double buffer of rounds 1-5 (experiment
# build, V1T_TN2_NST=2): kernel tests, the native step, the drop-in loop, an 8-rank share
O=$GRAFT_REPO_ROOT/gpurun_out/r06e
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -k "gemm_tn or batched_backward or fused_training or backward_fusions" > $O/pytest_tn.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest_tn.txt | tail -4
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "native ring3 : $(python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab_tn2.txt
  echo "native ring2 : $(V1T_LIB=libv1t_amd_exp.so V1T_TN2_NST=2 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab_tn2.txt
done
for i in 1 2; do
  echo "module ring3 : $(python bench.py --path module --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab_tn2.txt
  echo "module ring2 : $(V1T_LIB=libv1t_amd_exp.so V1T_TN2_NST=2 python bench.py --path module --no-cpu-baseline --no-pmc 2>/dev/null | line)" | tee -a $O/ab_tn2.txt
  echo "sim8 ring3   : $(SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_tn2.txt
  echo "sim8 ring2   : $(V1T_LIB=libv1t_amd_exp.so V1T_TN2_NST=2 SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_tn2.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm3 -- python3 $GRAFT_REPO_ROOT/bench.py --path module --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-pmc > /tmp/pm3.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pm3/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "tn" in r["Name"] or "dkv2" in r["Name"]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
