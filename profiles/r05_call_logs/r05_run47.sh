# round-5 GPU call 47: per-kernel times of the fused / unfused MLP backward under rocprofv3
O=$GRAFT_REPO_ROOT/gpurun_out/r05ah
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  export V1T_MLP_BWD_FUSE=$f
  rocprofv3 --kernel-trace --stats -d /tmp/prof_mb$f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --min-seconds 0 > /dev/null 2>&1
  echo "== V1T_MLP_BWD_FUSE=$f" | tee -a $O/kernels.txt
  python3 $GRAFT_REPO_ROOT/tools/prof_top.py $(ls /tmp/prof_mb$f/*/*kernel_stats.csv | head -1) 7 30 | grep -i "mlp_bwd\|lnbwd\|gemm_nt_kernel<4, 4\|total\|dkv2\|tn2" | tee -a $O/kernels.txt
done
for i in 1 2 3; do
  for f in 1 0; do
    echo "bench bwdfuse=$f: $(V1T_MLP_BWD_FUSE=$f python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/kernels.txt
  done
done
echo done
