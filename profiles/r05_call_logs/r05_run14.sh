# round-5 GPU call 14: second stream at a 2-GPU share (56 images); full suite on the current build
O=$GRAFT_REPO_ROOT/gpurun_out/r05n
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "N=2 default      : $(SIM_ONLY=2,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_n2.txt
  echo "N=2 dw_side=1    : $(V1T_DW_SIDE=1 SIM_ONLY=2,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_n2.txt
  echo "N=2 dw_side=1 wgs256: $(V1T_DW_SIDE=1 V1T_TN_WGS=256 SIM_ONLY=2,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_n2.txt
  echo "N=1 default      : $(SIM_ONLY=1,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_n2.txt
  echo "N=1 dw_side=1    : $(V1T_DW_SIDE=1 SIM_ONLY=1,0 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_n2.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
