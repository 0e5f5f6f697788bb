for i in 1 2 3; do
  echo -n "new: "; python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
  echo -n "old: "; V1T_KEEP_BF16_PLANES=1 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
done
