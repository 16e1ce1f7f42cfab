#!/usr/bin/env python3
"""Where does the spmm_max backward go?  grad_value only / grad_mat only / both,
config-3 size (2M x 2M, 20M edges, F = 128)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from paddle_sparse_amd import ops  # noqa: E402
from util import random_csr  # noqa: E402

M, nnz, F = 2_000_000, 20_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 128
row, rowptr, col, val = random_csr(M, M, nnz, 2)
rowptr_d, col_d, val_d = (torch.from_numpy(x).cuda() for x in (rowptr, col, val))
g = torch.Generator(device="cuda").manual_seed(1)
B = torch.randn(M, F, device="cuda", generator=g)
grad = torch.randn(M, F, device="cuda", generator=g)
out, arg = ops.spmm_max(rowptr_d, col_d, val_d, B)


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


print(f"F={F}")
print("both        ", ms(lambda: ops.spmm_minmax_bw(col_d, val_d, B, grad, arg, True, True)))
print("grad_value  ", ms(lambda: ops.spmm_minmax_bw(col_d, val_d, B, grad, arg, True, False)))
print("grad_mat    ", ms(lambda: ops.spmm_minmax_bw(col_d, val_d, B, grad, arg, False, True)))
print("zero 1 GB   ", ms(lambda: torch.zeros_like(B)))

# the atomic-free grad_mat over the CSC view
from paddle_sparse_amd import SparseStorage  # noqa: E402

st = SparseStorage(rowptr=rowptr_d, col=col_d, value=val_d, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
print("CSC build   ", ms(lambda: SparseStorage(rowptr=rowptr_d, col=col_d, value=val_d, sparse_sizes=(M, M),
                                               is_sorted=True, trust_data=True).csr2csc(), reps=5))
csr2csc = st.csr2csc()
print("edge tags   ", ms(lambda: ops.csc_edge_tags(st.rowptr(), st._row_in_csc_order(), csr2csc)))
tags = st._csc_edge_tags()


def csc(want_value):
    return ops.spmm_minmax_bw_csc(st.rowptr(), st.colptr(), st._row_in_csc_order(), csr2csc, tags, val_d, B, grad,
                                  arg, want_value=want_value, csc2csr=st.csc2csr())


gv_csc, gm_csc = csc(True)
print("CSC both    ", ms(lambda: csc(True)))
print("CSC grad_mat", ms(lambda: csc(False)))
gv_atomic, gm_atomic = ops.spmm_minmax_bw(col_d, val_d, B, grad, arg, True, True)
print("max |gm csc - atomic| =", float((gm_csc - gm_atomic).abs().max()), " max |gm| =", float(gm_atomic.abs().max()))
print("max |gv csc - row|    =", float((gv_csc - gv_atomic).abs().max()), " max |gv| =", float(gv_atomic.abs().max()))
