#!/bin/bash
# rocprofv3 passes for one round; run on the GPU box through gpurun:
#   gpurun -- 'bash tools/archive/profile.sh r01'
# Writes under gpurun_out/prof_<round>/ ; tools/summarize_profile.py turns it
# into the committed profiles/<round>_* summaries.
set -e -o pipefail
ROUND=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
echo "stats pass done"
for W in calib c3; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- python3 $REPO/tools/archive/pmc_workloads.py $W > $OUT/pmc_fetch_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- python3 $REPO/tools/archive/pmc_workloads.py $W > $OUT/pmc_write_$W.log 2>&1
  echo "pmc $W done"
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_c3 -- python3 $REPO/tools/archive/pmc_workloads.py c3 > $OUT/pmc_l2_c3.log 2>&1
echo "pmc l2 done"
find $OUT -name '*.csv' | head -50
