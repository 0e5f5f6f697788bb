# round-5 GPU calls 24-26: attention-P dropout at 16-bit rate resolution (A: one word per query and key pair, 16-bit halves; B: the same with DPP-shared words in the dK/dV producers; C: byte decisions against a per-tile dithered threshold) against the 8-bit build (libv1t_amd_old8.so)
O=$GRAFT_REPO_ROOT/gpurun_out/r05y
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_trajectory.py -x -q -m gpu -k "attention or dropout or lsa" 2>&1 | tail -5 | tee $O/pytest.log
for i in 1 2; do
  for lib in libv1t_amd.so libv1t_amd_old8.so; do
    for B in 112 14; do
      echo "B=$B $lib: $(ATTN_B=$B V1T_LIB=$lib python tools/attn_bench.py 20 2>/dev/null | grep "p=0.2544" | grep -i " fwd\|dkv_store\|bwd_dkv" | awk '{print $2, $3}' | tr '\n' ' ')" | tee -a $O/ab_drop16.txt
    done
  done
done
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_old8.so; do
    echo "bench $lib: $(V1T_LIB=$lib python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'])")" | tee -a $O/ab_drop16.txt
  done
done
for lib in libv1t_amd.so libv1t_amd_old8.so; do
  echo "sim8 $lib: $(V1T_LIB=$lib SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_drop16.txt
done
echo done
