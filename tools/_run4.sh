set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_misc_gpu.py tests/test_api_gpu.py tests/test_spmm_gpu.py tests/test_spmm_half_gpu.py tests/test_graph_capture_gpu.py tests/test_full_size_gpu.py tests/test_golden_gpu.py tests/test_cabi_client_gpu.py tests/test_distributed.py -x -q -m gpu > gpurun_out/t_bw.log 2>&1 || { tail -50 gpurun_out/t_bw.log; exit 1; }
tail -3 gpurun_out/t_bw.log
timeout -k 10 500 python tools/rmat_train_step.py 2>&1 | tee gpurun_out/rmat_train_step.txt
timeout -k 10 300 python tools/half_variants.py 2>&1 | tee gpurun_out/half_variants.txt
