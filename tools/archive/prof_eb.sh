#!/bin/bash
# rocprofv3 passes comparing SpMM variants on one graph: bash tools/archive/prof_eb.sh <tag> <c3|rmat> "<variants>" [op]
set -e -o pipefail
TAG=$1; G=$2; VARS=$3; OP=${4:-spmm_sum}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for V in $VARS; do
  OUT=$REPO/gpurun_out/prof_${TAG}_${G}_v$V
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/tools/archive/pmc_eb.py $G $V $OP 10 > $OUT/stats.log 2>&1
  i=0
  for CTRS in ${PASSES:-"FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"}; do
    i=$((i+1))
    rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc$i -- python3 $REPO/tools/archive/pmc_eb.py $G $V $OP 3 > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
  done
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/stats/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        name = r["Name"].replace("(anonymous namespace)::", "")[:60]
        if name.startswith("void at::") or "rocprim" in name: continue
        print(f'  {name:60s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:10.1f}')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "spmm" not in name: continue
        agg[name[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(" ", k)
    for c, v in sorted(d.items()):
        print(f"      {c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
done
