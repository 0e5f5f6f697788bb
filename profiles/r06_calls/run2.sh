# round-6 GPU call 2: the whole GPU suite on the product library without development switches (experiment build for the equality tests), bench, C4 / C5 / module lines
O=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -v "^ *[0-9.]*x " $O/pytest_gpu.txt | tail -40
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; python -c "import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
for p in module module-fused; do python bench.py --path $p --no-cpu-baseline > $O/bench_c2_$p.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench_c2_$p.json').read().strip().splitlines()[-1]); print('$p', d['value'], d['ms_per_step'])"; done
