set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_small_coalesce_gpu.py tests/test_sort_gpu.py tests/test_cabi_client_gpu.py tests/test_api_gpu.py -x -q -m gpu > gpurun_out/t_fused.log 2>&1 || { tail -40 gpurun_out/t_fused.log; exit 1; }
tail -3 gpurun_out/t_fused.log
python tests/perf/c1_small.py > gpurun_out/c1_small.txt 2>&1
cat gpurun_out/c1_small.txt
python tools/coalesce_sizes.py > gpurun_out/coalesce_sizes.txt 2>&1
cat gpurun_out/coalesce_sizes.txt
