#!/bin/bash
# A/B of the cross-lane folds on the GPU box: the library as built (DPP / permlane swaps), then rebuilt with every fold
# through ds_bpermute (-DPSA_SHFL_FOLDS), tools/fold_ab.py after each.   bash tools/fold_ab.sh > gpurun_out/r04_fold_ab.txt
set -e
python tools/fold_ab.py valu_folds
PSA_EXTRA_HIPCC_FLAGS=-DPSA_SHFL_FOLDS python -m paddle_sparse_amd.build --force > /dev/null
python tools/fold_ab.py shfl_folds
PSA_VBW_U8=1 python - <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from bench import event_ms, make_workload
from paddle_sparse_amd import ops
dev = torch.device("cuda", 0)
rowptr, col, val = make_workload(2_000_000, 2_000_000, 20_000_000, 128, 2, dev)
B = torch.randn(2_000_000, 128, device=dev); G = torch.randn(2_000_000, 128, device=dev)
fn = lambda: ops.spmm_value_bw(None, rowptr, col, B, G, "sum")
fn()
print(f"[shfl_folds, U = 8] config 3: spmm_value_bw alone: {event_ms(fn, 10):7.3f} ms")
PY
