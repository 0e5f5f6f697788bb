# round-5 GPU call 8: K-tile choice at a 4-GPU share, host time per step
O=$GRAFT_REPO_ROOT/gpurun_out/r05h
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "default : $(SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_bk.txt
  echo "bk64    : $(V1T_GEMM_BK=64 SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_bk.txt
done
echo "sim 8: $(SIM_STEPS=20 SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/host.txt
echo "sim 1: $(SIM_STEPS=10 SIM_ONLY=1,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/host.txt
python tools/sim_scaling.py 2>/dev/null | grep "^world" > $O/sim_scaling.txt
cat $O/sim_scaling.txt
bash $GRAFT_REPO_ROOT/tools/rank_census.sh 4 1 > $O/rank4.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
