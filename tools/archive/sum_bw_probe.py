#!/usr/bin/env python3
"""sum backward at config-3 size: the three-kernel form (value_bw + weight
gather + SpMM over CSC) against the one-pass CSC form."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from paddle_sparse_amd import SparseStorage, ops  # noqa: E402
from util import random_csr  # noqa: E402

M, nnz, F = 2_000_000, 20_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 128
row, rowptr, col, val = random_csr(M, M, nnz, 2)
rowptr_d, col_d, val_d = (torch.from_numpy(x).cuda() for x in (rowptr, col, val))
g = torch.Generator(device="cuda").manual_seed(1)
B = torch.randn(M, F, device="cuda", generator=g)
grad = torch.randn(M, F, device="cuda", generator=g)
st = SparseStorage(rowptr=rowptr_d, col=col_d, value=val_d, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
csr2csc, colptr, row_csc = st.csr2csc(), st.colptr(), st._row_in_csc_order()


def ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def three():
    gv = ops.spmm_value_bw(None, rowptr_d, col_d, B, grad, "sum")
    w = ops.transpose_weights(val_d, csr2csc, None, None, False)
    return gv, ops.spmm_sum(colptr, row_csc, w, grad)


print(f"F={F}")
print("value_bw + weights + SpMM over CSC", ms(three))
print("one CSC pass, grad_mat only       ", ms(lambda: ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val_d, B, grad, False)))
gv3, gm3 = three()
gv1, gm1 = ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val_d, B, grad, True, csc2csr=st.csc2csr())
print("max |gm diff|", float((gm1 - gm3).abs().max()), "max |gv diff|", float((gv1 - gv3).abs().max()),
      "max |gm|", float(gm3.abs().max()), "max |gv|", float(gv3.abs().max()))

csc2csr = st.csc2csr()
print("one CSC pass + gather to CSR order", ms(lambda: ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val_d, B, grad, True, csc2csr=csc2csr)))

# memory-path A/B: variant 17 = ordinary stores and an ordinary load of the column's own mat row
for v in (0, 17, 0, 17):
    ops.spmm_set_variant(v)
    t = ms(lambda: ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val_d, B, grad, True, csc2csr=csc2csr))
    print(f"variant {v}: one CSC pass + gather to CSR order {t:.3f} ms")
ops.spmm_set_variant(0)
