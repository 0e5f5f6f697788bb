# round-5 GPU call 33: forward attention with a branch-free K / V stage issue (every wave issues three K and three V pieces, the ragged check only in the
# peeled last stage) against the round-4 form (libv1t_amd_branchy.so = -DV1T_FWD_DMA_BRANCHY)
O=$GRAFT_REPO_ROOT/gpurun_out/r05ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_longseq.py -x -q -m gpu -k "attention or longseq" 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_branchy.so; do
    for B in 112 14 28; do
      echo "B=$B $lib: $(ATTN_B=$B V1T_LIB=$lib python tools/attn_bench.py 20 2>/dev/null | grep -i " fwd" | awk '{print $1, $2, $3}' | tr '\n' ' ')" | tee -a $O/ab_fwd_dma.txt
    done
  done
done
for i in 1 2 3; do
  for lib in libv1t_amd.so libv1t_amd_branchy.so; do
    echo "bench $lib: $(V1T_LIB=$lib python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'])")" | tee -a $O/ab_fwd_dma.txt
  done
done
for lib in libv1t_amd.so libv1t_amd_branchy.so; do
  echo "c5 $lib: $(V1T_LIB=$lib python bench.py --config c5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab_fwd_dma.txt
done
echo done
