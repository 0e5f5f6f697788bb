# round-6 GPU call 16: the dK/dV kernel and the dQ GEMM alone on the chip at 8 .. 28 images (round quantisation of small launches)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06o
python -c "import __graft_entry__ as g; g.build()" | tail -1
python tools/dkv2_rounds_probe.py 2>&1 | tee gpurun_out/r06o/dkv2_rounds.txt
