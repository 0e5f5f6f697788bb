# round-5 GPU call 28b: Weyl-dithered byte thresholds (5 scalar instructions per tile, SGPR operand in the SDWA compare) against the 8-bit build
O=$GRAFT_REPO_ROOT/gpurun_out/r05y2
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_old8.so; do
    echo "B=112 $lib: $(ATTN_B=112 V1T_LIB=$lib python tools/attn_bench.py 20 2>/dev/null | grep "p=0.2544" | grep -i " fwd\|dkv_store\|bwd_dkv" | awk '{print $2, $3}' | tr '\n' ' ')" | tee -a $O/ab_drop16.txt
  done
done
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_old8.so; do
    echo "bench $lib: $(V1T_LIB=$lib python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'])")" | tee -a $O/ab_drop16.txt
  done
done
bash tools/r05_run29.sh
