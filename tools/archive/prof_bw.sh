#!/bin/bash
# PMC passes of the min/max backward over the CSC view on R-MAT 21: bash tools/archive/prof_bw.sh "<relabel values>"
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for R in ${1:-0 1}; do
  OUT=$REPO/gpurun_out/prof_bw_$R
  mkdir -p $OUT
  i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc$i -- python3 $REPO/tools/archive/rmat_bw_prof.py $R 1 > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
  done
  echo "== relabel $R (hub copies on)"
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmm_fused_kernel<4, 32, 0, 4, 1>" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"      {c:40s} n={len(v):3d} mean={sum(v)/len(v):18.1f}")
PY
done
