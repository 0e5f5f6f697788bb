# round-5 GPU call 12: full suite (new optimizer test, guarded ablations build), c5, quick numbers
O=$GRAFT_REPO_ROOT/gpurun_out/r05l
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
grep -n "FusedAdamW.for_model" $O/pytest.log | head -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
