cd /tmp && export TMPDIR=/tmp
for V in 0 1 2 3; do
  V1T_DBG_GEMM=$V V1T_DBG_NOCOLSUM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$V -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== dbg $V"; grep -h "gemm_nt_kernel<4, 4>" /tmp/p_$V/*/*kernel_stats.csv | cut -d, -f1-4
done
