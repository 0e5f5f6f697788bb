# round-5 GPU call 31: where the host spends its 24 ms per step of the per-mouse loop (cProfile over bench.py --path module)
O=$GRAFT_REPO_ROOT/gpurun_out/r05aa
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m cProfile -o $O/module.prof bench.py --path module --steps 20 --warmup 3 --min-seconds 0 --no-cpu-baseline > $O/line.json 2> $O/err.txt
python - <<'P' > $O/host_profile.txt
import pstats, os
p = pstats.Stats(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05aa/module.prof")
p.sort_stats("cumulative").print_stats(70)
p.sort_stats("tottime").print_stats(45)
P
tail -3 $O/err.txt
head -c 600 $O/line.json
echo done
