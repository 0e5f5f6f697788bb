# usage (GPU box): bash tools/ab_env.sh VAR [rounds]  -> bench.py step time with and without the dev switch VAR=1, interleaved
VAR=$1; N=${2:-3}
for i in $(seq $N); do
  echo -n "default : "; python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
  echo -n "$VAR=1: "; env $VAR=1 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
done
