# round-6 GPU call 3: the failing sharp-regime checks with every tensor measured; A/B of the two scheduling changes (weight-gradient group handed over
# behind the dO GEMM; the tails' side work released at the first attention backward); two half-batch steps on two streams; counters available
O=$GRAFT_REPO_ROOT/gpurun_out/r06c
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trajectory.py -m gpu -q -k "extreme or replayed or direct" > $O/pytest_sharp.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest_sharp.txt | tail -15
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])"; }
for i in 1 2 3; do
  echo "A default(tails late)      : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_sched.txt
  echo "B tails early (round 5)    : $(V1T_TAILS_EARLY=1 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_sched.txt
  echo "C exp lib, flush late      : $(V1T_LIB=libv1t_amd_exp.so V1T_DW_FLUSH_LATE=1 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_sched.txt
  echo "D exp lib, default         : $(V1T_LIB=libv1t_amd_exp.so python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/ab_sched.txt
done
python tools/dual_stream_probe.py 2>&1 | tail -6 | tee $O/dual_stream.txt
V1T_DW_SIDE=0 python tools/dual_stream_probe.py 2>&1 | tail -6 | tee $O/dual_stream_noside.txt
(rocprofv3-avail list 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) > $O/counters_avail.txt 2>&1; grep -i -o "TCC_EA[A-Z0-9_]*\|TCC_[A-Z0-9_]*REQ[A-Z0-9_]*\|MALL[A-Z0-9_]*" $O/counters_avail.txt | sort -u | head -80
