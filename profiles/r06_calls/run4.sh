# round-6 GPU call 4: the dK/dV kernel with exact scores (K no longer re-rounded as bf16(c k)): the trained-regime checks again, the step's time
# (3 runs), read-request sizes of every kernel (FETCH_SIZE calibration)
O=$GRAFT_REPO_ROOT/gpurun_out/r06d
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trajectory.py tests/test_gpu_parity.py -m gpu -q -k "extreme or replayed or direct or attention_forward_backward or dropout_replayed or g14 or golden" > $O/pytest_sharp.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest_sharp.txt | tail -8
python - <<'PY'
import json
m = json.load(open("gpurun_out/parity_margins.json"))
for k, v in sorted(m.items(), key=lambda kv: -kv[1]["ratio"])[:14]:
    print(f"{v['ratio']:.3f} {v['err']:.3e} {v['bound']:.1e} {k}")
PY
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])"; }
for i in 1 2 3; do echo "bench: $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/bench3.txt; done
RND=r06 bash tools/pmc_readsize.sh > $O/readsize.txt 2>&1; tail -30 $O/readsize.txt
