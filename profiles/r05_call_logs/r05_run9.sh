# round-5 GPU call 9: second stream on / off at a 4-GPU share (28 images) and at 16 images (module path)
O=$GRAFT_REPO_ROOT/gpurun_out/r05i
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for side in 1 0; do
    echo "N=4 dw_side=$side: $(V1T_DW_SIDE=$side SIM_ONLY=4,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_side.txt
  done
  for side in 1 0; do
    echo "N=8 dw_side=$side: $(V1T_DW_SIDE=$side SIM_ONLY=8,1 python tools/sim_scaling.py 2>/dev/null | tail -1)" | tee -a $O/ab_side.txt
  done
done
for side in 1 0; do
  V1T_DW_SIDE=$side python bench.py --path module --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/m.json
  python - <<PY
import json
d=json.load(open("$O/m.json")); print("module dw_side=$side", d["value"], d["ms_per_step"])
PY
done | tee -a $O/ab_side.txt
V1T_DW_SIDE=0 bash $GRAFT_REPO_ROOT/tools/rank_census.sh 4 1 > $O/rank4_noside.txt 2>&1
