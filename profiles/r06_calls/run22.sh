# round-6 GPU call 22: is delta = rowsum(dO o O) more accurate from the STORED bf16 dO (consistent with the dP the kernel computes from it: attn_delta2_kernel,
# experiment build V1T_DELTA_UNFUSED=1) than from the dO GEMM's fp32 accumulators (row-dot epilogue, product)? trained-regime replay tests, margins of both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06t
python -c "import __graft_entry__ as g; g.build()" | tail -1
for mode in product unfused; do
  if [ $mode = unfused ]; then export V1T_LIB=libv1t_amd_exp.so V1T_DELTA_UNFUSED=1; fi
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py tests/test_gpu_kernels.py -m gpu -q -k "dropout_replayed or replayed_masks or extreme or g14" > gpurun_out/r06t/pytest_$mode.txt 2>&1; grep -v "^ *[0-9.]*x " gpurun_out/r06t/pytest_$mode.txt | tail -2 | cut -c1-200
  cp gpurun_out/parity_margins.json gpurun_out/r06t/margins_$mode.json
done
python - <<'PY'
import json
a = json.load(open("gpurun_out/r06t/margins_product.json")); b = json.load(open("gpurun_out/r06t/margins_unfused.json"))
rows = sorted(a.items(), key=lambda kv: -kv[1]["ratio"])[:16]
for k, v in rows:
    print(f"{v['ratio']:.3f} -> {b.get(k, {}).get('ratio', float('nan')):.3f}  {k[:110]}")
PY
