#!/usr/bin/env python3
"""Workload for --pmc passes on the backward kernels at config-3 size: three
calls each of the one-pass CSC sum backward, the one-pass CSC max backward and
spmm_value_bw."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_workload  # noqa: E402
from paddle_sparse_amd import SparseStorage, ops  # noqa: E402

dev = torch.device("cuda", 0)
M = N = 2_000_000
F = 128
rowptr, col, val = make_workload(M, N, 20_000_000, F, 2, dev)
g = torch.Generator(device=dev).manual_seed(1)
B = torch.randn(N, F, generator=g, device=dev)
G = torch.randn(M, F, generator=g, device=dev)
st = SparseStorage(rowptr=rowptr, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
csr2csc, colptr, row_csc, inv, tags = st.csr2csc(), st.colptr(), st._row_in_csc_order(), st.csc2csr(), st._csc_edge_tags()
# what autograd leaves behind the forward on config 3 (no row above 128 entries): `out` and the one-byte
# row-local arg_out only (matmul.py) — the M_MASK instantiation that reads bytes and never arg_out
out, _, arg_bytes = ops._spmm("max", rowptr, col, val, B, want_arg_bytes=1, want_arg=False)
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(reps):
    ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=inv)
    ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, tags, val, B, G, None, csc2csr=inv, arg_bytes=arg_bytes)
    ops.spmm_value_bw(None, rowptr, col, B, G, "sum")
torch.cuda.synchronize()
print("done")
