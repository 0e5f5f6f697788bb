# round-5 GPU call 46: the MLP branch backward as one launch (mlp_bwd_kernel) against gemm_nt<EPI_DGELU> + gemm_lnbwd (V1T_MLP_BWD_FUSE=0)
O=$GRAFT_REPO_ROOT/gpurun_out/r05ag
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_mlp" 2>&1 | tail -6 | tee $O/pytest.log
for i in 1 2; do
  for f in 1 0; do
    echo "bench bwdfuse=$f: $(V1T_MLP_BWD_FUSE=$f python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])")" | tee -a $O/ab_mlp_bwd.txt
  done
done
for f in 1 0 2; do
  echo "sim4 bwdfuse=$f: $(V1T_MLP_BWD_FUSE=$f SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_mlp_bwd.txt
  echo "sim8 bwdfuse=$f: $(V1T_MLP_BWD_FUSE=$f SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_mlp_bwd.txt
done
echo done
