# round-5 GPU call 29: fused criterion / ELU1 backward in the per-mouse loop; host enqueue time of the module path
O=$GRAFT_REPO_ROOT/gpurun_out/r05z
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for path in module module-fused; do
    echo "$path: $(python bench.py --path $path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['windows'].get('host_enqueue_ms_per_step'))")" | tee -a $O/module.txt
  done
done
echo "native: $(python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['windows'].get('host_enqueue_ms_per_step'))")" | tee -a $O/module.txt
echo done
