# round-5 GPU call 35: the MLP branch forward as one launch (mlp_fwd_kernel) against LN2+FC1 and FC2 as two (V1T_MLP_FUSE=0)
O=$GRAFT_REPO_ROOT/gpurun_out/r05ad
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2; do
  for f in 1 2 0; do
    echo "bench fuse=$f: $(V1T_MLP_FUSE=$((f>0)) V1T_MLP_NBLK=$f python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])")" | tee -a $O/ab_mlp.txt
  done
done
for i in 1; do
  for f in 1 2 0; do
    echo "sim8 fuse=$f: $(V1T_MLP_FUSE=$((f>0)) V1T_MLP_NBLK=$f SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_mlp.txt
    echo "sim4 fuse=$f: $(V1T_MLP_FUSE=$((f>0)) V1T_MLP_NBLK=$f SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_mlp.txt
  done
done
for f in 1 2 0; do
  echo "module fuse=$f: $(V1T_MLP_FUSE=$((f>0)) V1T_MLP_NBLK=$f python bench.py --path module 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab_mlp.txt
  echo "c5 fuse=$f: $(V1T_MLP_FUSE=$((f>0)) V1T_MLP_NBLK=$f python bench.py --config c5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab_mlp.txt
done
echo done
