#!/bin/bash
# rocprofv3 passes of round 4 (on the GPU box): bash tools/profile_r04.sh <what>
#   bench      --kernel-trace --stats of the headline bench.py run (no extra legs) -> gpurun_out/r04_kernel_stats.csv
#   half       two --pmc passes (instruction counts, wave cycles) over tools/half_train_step_probe.py: the half-width
#              forward and the passes over the CSC view, for profiles/r04_pmc_half.json (before: profiles/r03_pmc_half.json)
#   traffic    FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes over the headline bench command -> profiles/r04_pmc.json, traffic.json
#   value_bw   stats + FETCH_SIZE / WRITE_SIZE passes of tools/pmc_backward.py (spmm_value_bw among its kernels)
# Counters are collected in runs of their own (never together with a trace), as gpurun requires.
set -e -o pipefail
WHAT=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
case $WHAT in
  bench)
    bash tools/prof_stats.sh r04_bench $REPO/bench.py --steps 40 --warmup 5 --no-cpu --no-extra | tee gpurun_out/r04_bench_stats.txt
    cp gpurun_out/stats_r04_bench/*/*_kernel_stats.csv gpurun_out/r04_kernel_stats.csv ;;
  half)
    for c in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
      tag=r04_half_$(echo $c | tr ' ' '_' | cut -c1-24)
      bash tools/prof_pmc.sh $tag "$c" $REPO/tools/half_train_step_probe.py | grep -A6 "spmm_half_row_kernel<[a-z_:]*BF16, 16, 0, 4\|spmm_half_csc_bw_kernel<[a-z_:]*BF16, 16, 4, true" | tee -a gpurun_out/r04_half_pmc.txt
    done ;;
  traffic)
    # HBM bytes of the headline kernel: separate FETCH_SIZE / WRITE_SIZE / TCC passes over the bench command
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      tag=r04_traffic_$(echo $c | tr ' ' '_' | cut -c1-12)
      bash tools/prof_pmc.sh $tag "$c" $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-extra | grep -A3 "spmm_fused_kernel<4, 32, 0, 4, 0" | tee -a gpurun_out/r04_traffic_pmc.txt
    done ;;
  value_bw)
    bash tools/prof_stats.sh r04_bw $REPO/tools/pmc_backward.py 10 | tee gpurun_out/r04_bw_stats.txt
    for c in FETCH_SIZE WRITE_SIZE; do
      bash tools/prof_pmc.sh r04_bw_$c "$c" $REPO/tools/pmc_backward.py 3 | tee gpurun_out/r04_bw_$c.txt
    done ;;
esac
