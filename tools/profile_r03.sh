#!/bin/bash
# rocprofv3 passes of round 3 (on the GPU box): bash tools/profile_r03.sh <what>
#   backward   stats + FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes of the CSC-view backward kernels at config 3
#   breadth    stats of a bench.py run including c3_other_ops / power_law
#   bench      stats of the headline bench.py run (no extra legs)
set -e -o pipefail
WHAT=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
case $WHAT in
  backward)
    bash tools/prof_stats.sh r03_bw $REPO/tools/pmc_backward.py 10 | tee gpurun_out/r03_bw_stats.txt
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
      tag=r03_bw_$(echo $c | tr ' ' '_' | cut -c1-24)
      bash tools/prof_pmc.sh $tag "$c" $REPO/tools/pmc_backward.py 3 | tee gpurun_out/$tag.txt
    done ;;
  breadth)
    bash tools/prof_stats.sh r03_breadth $REPO/bench.py --steps 20 --warmup 5 --no-cpu | tee gpurun_out/r03_breadth_stats.txt
    cp gpurun_out/stats_r03_breadth/*/*_kernel_stats.csv gpurun_out/r03_kernel_stats_breadth.csv ;;
  traffic)
    # HBM bytes of the headline kernel: separate FETCH_SIZE / WRITE_SIZE / TCC passes over the bench command
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      tag=r03_traffic_$(echo $c | tr ' ' '_' | cut -c1-12)
      bash tools/prof_pmc.sh $tag "$c" $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-extra | grep -A3 "spmm_fused_kernel<4, 32, 0, 4, 0" | tee -a gpurun_out/r03_traffic_pmc.txt
    done ;;
  sort)
    bash tools/prof_stats.sh r03_sort $REPO/tools/pmc_sort.py | tee gpurun_out/r03_sort_stats.txt
    for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM"; do
      tag=r03_sort_$(echo $c | tr ' ' '_' | cut -c1-28)
      bash tools/prof_pmc.sh $tag "$c" $REPO/tools/pmc_sort.py | grep -A8 "os_pass_kernel<false, false" | tee -a gpurun_out/r03_sort_pmc.txt
    done ;;
  bench)
    bash tools/prof_stats.sh r03_bench $REPO/bench.py --steps 40 --warmup 5 --no-cpu --no-extra | tee gpurun_out/r03_bench_stats.txt
    cp gpurun_out/stats_r03_bench/*/*_kernel_stats.csv gpurun_out/r03_kernel_stats.csv ;;
esac
