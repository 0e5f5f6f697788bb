# round-5 GPU call 13: forward attention with 4 full + 5 half row blocks at 14 images
O=$GRAFT_REPO_ROOT/gpurun_out/r05m
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
for B in 14 16 28; do
  for sp in 1 0 1 0; do
    echo "B=$B split=$sp: $(ATTN_B=$B V1T_FWD_SPLIT=$sp python tools/attn_bench.py 20 2>/dev/null | grep -i fwd | tr '\n' ' ')" | tee -a $O/ab_split.txt
  done
done
for i in 1 2; do
  for sp in 1 0; do
    echo "split=$sp: $(V1T_FWD_SPLIT=$sp SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_split_sim.txt
  done
done
python tools/sim_scaling.py 2>/dev/null | grep "^world" | tee $O/sim_scaling.txt
