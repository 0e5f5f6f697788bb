# round-5 GPU call 19: ln_gemm column split / shape at a 4-GPU share (28 images) and at 14 / 16 images
O=$GRAFT_REPO_ROOT/gpurun_out/r05r
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in "default:" "split2:V1T_LNG_SPLIT=2" "split3:V1T_LNG_SPLIT=3" "shape2:V1T_LNG_SHAPE=2"; do
    name=${v%%:*}; envs=${v#*:}
    echo "N=4 $name: $(env $envs SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_lng.txt
  done
done
for v in "default:" "split1:V1T_LNG_SPLIT=1" "split3:V1T_LNG_SPLIT=3" "split4:V1T_LNG_SPLIT=4"; do
  name=${v%%:*}; envs=${v#*:}
  echo "N=8 $name: $(env $envs SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_lng.txt
  echo "N=8 $name: $(env $envs SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_lng.txt
done
