# round-6 GPU call 21: dropout-replay parity in the trained-weights regime for the LSA / fused-recompute backward and the production kernels (B = 2)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06s
python -c "import __graft_entry__ as g; g.build()" | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dropout_replayed" > gpurun_out/r06s/pytest.txt 2>&1; grep -v "^ *[0-9.]*x " gpurun_out/r06s/pytest.txt | tail -6 | cut -c1-300
python - <<'PY'
import json
m = json.load(open("gpurun_out/parity_margins.json"))
for k, v in sorted(m.items(), key=lambda kv: -kv[1]["ratio"])[:8]:
    print(f"{v['ratio']:.3f} {v['err']:.3e} {v['bound']:.1e} {k[:120]}")
PY
