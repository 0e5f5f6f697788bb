# round-6 GPU call 20: SQ counters (MFMA busy, waits, VALU) of every kernel of the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" | tail -1
RND=r06 bash tools/pmc_sq_all.sh 2>&1 | tail -30
