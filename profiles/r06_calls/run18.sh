# round-6 GPU call 18: input gradient in train mode (patch dropout replayed) + the whole GPU suite once more on the final build
O=$GRAFT_REPO_ROOT/gpurun_out/r06q
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest_gpu.txt | tail -4
grep "input gradient" gpurun_out/parity_margins.json | head -0
python - <<'PY'
import json
m = json.load(open("gpurun_out/parity_margins.json"))
for k, v in m.items():
    if "dropout_replayed_in_oracle" in k and "input gradient" in k:
        print(f"{v['ratio']:.3f} {v['err']:.3e} {k}")
PY
