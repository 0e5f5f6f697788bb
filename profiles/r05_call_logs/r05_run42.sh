# round-5 GPU call 42: m-chunk target of the weight-gradient GEMMs at 112 images (V1T_TN_WGS), and the second stream at 112 images on the current build
O=$GRAFT_REPO_ROOT/gpurun_out/r05ae
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for w in 512 256 384 768 1024; do
    echo "tn_wgs=$w: $(V1T_TN_WGS=$w python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $O/tn_wgs.txt
  done
  echo "dw_side=1: $(V1T_DW_SIDE=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])")" | tee -a $O/tn_wgs.txt
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_tn -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --min-seconds 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_top.py $(ls /tmp/prof_tn/*/*kernel_stats.csv | head -1) 7 30 | grep -i "tn2\|tn_reduce\|total" | tee -a $O/tn_wgs.txt
echo done
