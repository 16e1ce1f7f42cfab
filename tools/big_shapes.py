#!/usr/bin/env python3
"""Shapes past 2^31 ELEMENTS (not bytes) through the SpMM forward and both CSC
backward passes: 20M x 20M, 40M edges, K = 128 -> every dense operand has
2.56e9 elements.  Catches 32-bit index arithmetic that the test-suite sizes
cannot reach.  Checked against rocSPARSE (torch.sparse.mm) for the forward and
against the three-kernel / atomic forms for the backward passes."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import SparseStorage, ops  # noqa: E402

M = N = 20_000_000
NNZ, K = 40_000_000, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(9)
key = torch.unique(torch.randint(0, M * N, (NNZ,), generator=g, device=dev))
row, col = torch.div(key, N, rounding_mode="floor"), key % N
del key
val = torch.randn(row.numel(), generator=g, device=dev)
B = torch.randn(N, K, generator=g, device=dev)
G = torch.randn(M, K, generator=g, device=dev)
st = SparseStorage(row=row, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
rowptr = st.rowptr()
print(f"{M} x {N}, nnz {row.numel()}, K {K}: dense operands hold {M * K:.3e} elements (> 2^31 = {2**31:.3e})")

out = ops.spmm_sum(rowptr, col, val, B)
ref = torch.sparse.mm(torch.sparse_csr_tensor(rowptr, col, val, size=(M, N)), B)
scale = torch.sparse.mm(torch.sparse_csr_tensor(rowptr, col, val.abs(), size=(M, N)), B.abs())
print("spmm_sum vs rocSPARSE: max err / sum|terms| =", float(((out - ref).abs() / (scale + 1e-30)).max()))
tail = slice(M - 1000, M)
print("  last rows non-trivial:", bool(out[tail].abs().sum() > 0), " first rows:", bool(out[:1000].abs().sum() > 0))
del ref, scale

csr2csc, colptr, row_csc, csc2csr = st.csr2csc(), st.colptr(), st._row_in_csc_order(), st.csc2csr()
gv1, gm1 = ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=csc2csr)
gv3 = ops.spmm_value_bw(None, rowptr, col, B, G, "sum")
gm3 = ops.spmm_sum(colptr, row_csc, ops.transpose_weights(val, csr2csc, None, None, False), G)
print("sum backward, one CSC pass vs three kernels: max |d grad_mat| =", float((gm1 - gm3).abs().max()),
      " max |d grad_value| =", float((gv1 - gv3).abs().max()))
del gm1, gm3, gv1, gv3

out, arg = ops.spmm_max(rowptr, col, val, B)
gv_c, gm_c = ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, st._csc_edge_tags(), val, B, G, arg,
                                    csc2csr=csc2csr)
gv_a, gm_a = ops.spmm_minmax_bw(col, val, B, G, arg)
print("max backward, one CSC pass vs atomics: max |d grad_mat| =", float((gm_c - gm_a).abs().max()),
      " max |d grad_value| =", float((gv_c - gv_a).abs().max()), " (|grad_mat| max", float(gm_a.abs().max()), ")")
torch.cuda.synchronize()
print("peak HBM in use: %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
