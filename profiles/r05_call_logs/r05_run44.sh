# round-5 GPU call 44: second stream at 112 images: m-chunk target and queue priority
O=$GRAFT_REPO_ROOT/gpurun_out/r05af
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for w in 512 256 384 768; do
    echo "tn_wgs=$w: $(V1T_TN_WGS=$w python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])")" | tee -a $O/side112.txt
  done
  echo "prio=low: $(V1T_DW_PRIO=low python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])")" | tee -a $O/side112.txt
done
echo done
