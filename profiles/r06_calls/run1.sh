# round-6 GPU call 1: the new parity tests (trained-regime golden G14, input gradient, amp, saturation, extreme-score backward, replay cases),
# the bench line, counters for every kernel of the step
O=$GRAFT_REPO_ROOT/gpurun_out/r06a
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_trajectory.py -m gpu -q -x -k "golden or input_gradient or amp or saturation or extreme or replayed or resize or unsupported" > $O/pytest_new.txt 2>&1; tail -70 $O/pytest_new.txt
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 1500 $O/bench_c2.json
RND=r06 bash tools/pmc_all.sh > $O/pmc_all.txt 2>&1; tail -60 $O/pmc_all.txt
