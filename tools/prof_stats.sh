#!/bin/bash
# usage (on the GPU box): bash tools/prof_stats.sh <tag> <python script + args...>
# rocprofv3 --kernel-trace --stats of one script; prints the top kernels (names trimmed).
set -e -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    name = r["Name"].replace("(anonymous namespace)::", "")[:70]
    print(f'{name:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:10.1f} pct={r["Percentage"]}')
PY
