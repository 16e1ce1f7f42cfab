#!/bin/bash
# Per-L2-channel request counts of the edge-balanced SpMM on R-MAT as generated and with relabelled columns:
#   bash tools/archive/tcc_channels.sh   (on the GPU box)
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for G in rmat rmat_relabel; do
  OUT=$REPO/gpurun_out/tcc_$G
  mkdir -p $OUT
  rocprofv3 --pmc TCC_REQ TCC_EA0_RDREQ --output-format csv -d $OUT/p1 -- python3 $REPO/tools/archive/pmc_eb.py $G 30 spmm_sum 2 > $OUT/p1.log 2>&1 || echo fail
  rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_BUSY --output-format csv -d $OUT/p2 -- python3 $REPO/tools/archive/pmc_eb.py $G 30 spmm_sum 2 > $OUT/p2.log 2>&1 || echo fail
  python3 - "$OUT" "$G" <<'PY'
import csv, glob, sys, collections
out, g = sys.argv[1], sys.argv[2]
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "spmm_eb_kernel" in r["Kernel_Name"]]
    if not rows: continue
    print(g, "columns:", list(rows[0].keys())[:14])
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append(r)
    for c, rs in by.items():
        vals = [float(r["Counter_Value"]) for r in rs]
        print(f"  {c}: rows={len(rs)} sum={sum(vals):.0f} min={min(vals):.0f} max={max(vals):.0f} max/mean={max(vals)/(sum(vals)/len(vals)):.2f}")
PY
done
