set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_small_coalesce_gpu.py tests/test_sort_gpu.py tests/test_cabi_client_gpu.py tests/test_api_gpu.py tests/test_pipeline_gpu.py tests/test_graph_capture_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/t_chain.log 2>&1 || { tail -40 gpurun_out/t_chain.log; exit 1; }
tail -3 gpurun_out/t_chain.log
python tools/coalesce_sizes.py > gpurun_out/coalesce_sizes.txt 2>&1
cat gpurun_out/coalesce_sizes.txt
for n in 100000 1000000; do echo "== $n"; bash tools/prof_stats.sh coal_$n $GRAFT_REPO_ROOT/tools/coalesce_prof.py $n 50; done > gpurun_out/coalesce_kernels.txt 2>&1
cat gpurun_out/coalesce_kernels.txt
