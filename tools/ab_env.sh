# usage (GPU box): bash tools/ab_env.sh VAR [rounds]  -> bench.py step time of the product library against the experiment build (libv1t_amd_exp.so) with the dev switch VAR=1, interleaved
VAR=$1; N=${2:-3}
for i in $(seq $N); do
  echo -n "default : "; python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
  echo -n "$VAR=1: "; env V1T_LIB=libv1t_amd_exp.so $VAR=1 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"
done
