# round-6 GPU call 7: the whole GPU suite on the final build, then every profile of the round (tools/refresh_profiles.sh + counters of every kernel)
O=$GRAFT_REPO_ROOT/gpurun_out/r06g
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest_gpu.txt | tail -6
cp gpurun_out/parity_margins.json $O/parity_margins_full.json
RND=r06 bash tools/refresh_profiles.sh > $O/refresh.txt 2>&1; tail -5 $O/refresh.txt
RND=r06 bash tools/pmc_all.sh > $O/pmc_all.txt 2>&1; tail -3 $O/pmc_all.txt
