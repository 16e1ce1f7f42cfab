#!/bin/bash
# usage (on the GPU box): bash tools/prof_pmc.sh <tag> "<CTR1 CTR2 ...>" <python script + args...>
# one rocprofv3 --pmc pass (counters only, no tracing); prints per-kernel means.
set -e -o pipefail
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if name.startswith("void at::") or "rocprim" in name or name.startswith("__amd") or name.startswith("at::"):
        continue
    agg[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
