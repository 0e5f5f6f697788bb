# round-6 GPU call 13: readout forward / parameter-gradient kernels with 4 images' sample positions and row reads in flight (was: one image at a
# time, 196 / 291 us per step): tests, the kernels' durations under rocprofv3, the step
O=$GRAFT_REPO_ROOT/gpurun_out/r06m
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "readout or golden or native_step_equals or fused_training" > $O/pytest.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest.txt | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do echo "native : $(python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/bench.txt; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > /tmp/pr.log 2>&1
python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/pr/*/*kernel_stats.csv")[0])):
    if "readout" in r["Name"] or "elu1" in r["Name"]:
        print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
