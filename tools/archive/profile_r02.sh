#!/bin/bash
# rocprofv3 passes of round 2 (on the GPU box): bash tools/archive/profile_r02.sh
# stats of bench.py, PMC traffic of the headline kernel (calibration + config 3), PMC of the
# edge-range forward with the hot copy on R-MAT, PMC of one sort pass.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/bench_under_rocprof.log 2>&1
echo "stats pass done"
for W in calib c3; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- python3 $REPO/tools/archive/pmc_workloads.py $W > $OUT/pmc_fetch_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- python3 $REPO/tools/archive/pmc_workloads.py $W > $OUT/pmc_write_$W.log 2>&1
  echo "pmc $W done"
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_c3 -- python3 $REPO/tools/archive/pmc_workloads.py c3 > $OUT/pmc_l2_c3.log 2>&1
echo "pmc l2 done"
