# round-6 GPU call 23: 2000 optimizer steps of the headline configuration on fixed synthetic batches (race / divergence check of the final build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" | tail -1
python tools/train_sanity.py 2000 2>&1 | awk 'NR % 10 == 1 || /299|1999|finite|Error|error/' | tail -14
