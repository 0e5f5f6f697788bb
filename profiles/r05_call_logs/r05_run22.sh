# round-5 GPU call 22: LPT / cover choice of the forward attention at 56 images (a 2-GPU share)
O=$GRAFT_REPO_ROOT/gpurun_out/r05t
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for mx in 1152 2048; do
    echo "B=56 lpt_max=$mx: $(ATTN_B=56 V1T_FWD_LPT_MAX=$mx python tools/attn_bench.py 20 2>/dev/null | grep -i " fwd" | tr '\n' ' ')" | tee -a $O/ab_lpt56.txt
  done
done
for i in 1 2; do
  for mx in 1152 2048; do
    echo "N=2 lpt_max=$mx: $(V1T_FWD_LPT_MAX=$mx SIM_ONLY=2,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_lpt56.txt
  done
done
