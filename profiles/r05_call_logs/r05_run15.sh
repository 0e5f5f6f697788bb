# round-5 GPU call 15: second stream (grouped hand-over) at the full 112-image step
O=$GRAFT_REPO_ROOT/gpurun_out/r05o
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for cfg in "default:" "side1:V1T_DW_SIDE=1" "side1_wgs256:V1T_DW_SIDE=1 V1T_TN_WGS=256" "side1_wgs384:V1T_DW_SIDE=1 V1T_TN_WGS=384"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    env $envs python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/b.json
    python - <<PY | tee -a $O/ab_side_n1.txt
import json
d=json.load(open("$O/b.json")); print("$name", d["value"], d["ms_per_step"], d["roofline"]["avg_ms"])
PY
  done
done
