# round-6 GPU call 14: 300 optimizer steps on fixed synthetic batches through the native step and through the reference's loop (torch.optim.AdamW / FusedAdamW.for_model)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" | tail -1
( echo "== python tools/train_sanity.py 300 (native step)"; python tools/train_sanity.py 300 2>&1 | tail -20
  echo "== python tools/train_sanity_module.py 300 (reference loop, torch.optim.AdamW)"; python tools/train_sanity_module.py 300 2>&1 | tail -20
  echo "== python tools/train_sanity_module.py 300 fused (reference loop, FusedAdamW.for_model)"; python tools/train_sanity_module.py 300 fused 2>&1 | tail -20 ) > gpurun_out/r06_train_sanity.txt 2>&1
tail -4 gpurun_out/r06_train_sanity.txt
