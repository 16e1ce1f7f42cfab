set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/t_full.log 2>&1 || { tail -40 gpurun_out/t_full.log; exit 1; }
tail -3 gpurun_out/t_full.log
python tools/coalesce_sizes.py > gpurun_out/coalesce_sizes.txt 2>&1
cat gpurun_out/coalesce_sizes.txt
python tests/perf/c1_small.py > gpurun_out/c1_small.txt 2>&1
cat gpurun_out/c1_small.txt
for n in 10000 100000 1000000; do echo "== $n"; bash tools/prof_stats.sh coal_$n $GRAFT_REPO_ROOT/tools/coalesce_prof.py $n 50; done > gpurun_out/coalesce_kernels.txt 2>&1
cat gpurun_out/coalesce_kernels.txt
python tools/sort_tiles.py > gpurun_out/sort_tiles.txt 2>&1
cat gpurun_out/sort_tiles.txt
