# round-6 GPU call 24: input gradient with a frozen model (the MEI setting)
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "input_gradient" 2>&1 | grep -v "^ *[0-9.]*x " | tail -8 | cut -c1-250
