# round-5 GPU call 37: fused MLP forward adopted above 256 row tiles: its equality test, then the suite + full refresh
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_mlp" 2>&1 | tail -5
bash tools/r05_run16.sh
