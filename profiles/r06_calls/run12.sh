# round-6 GPU call 12: the four weight-gradient GEMM shapes alone on the chip at 16 / 14 / 112 images over the m-chunk target
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06l
python -c "import __graft_entry__ as g; g.build()" | tail -1
python tools/tn_small_bench.py 2>&1 | tee gpurun_out/r06l/tn_small.txt
echo "--- NST=2 (experiment build)"; V1T_LIB=libv1t_amd_exp.so V1T_TN2_NST=2 python tools/tn_small_bench.py 2>&1 | head -8 | tee gpurun_out/r06l/tn_small_nst2.txt
