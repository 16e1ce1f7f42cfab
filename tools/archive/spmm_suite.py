#!/usr/bin/env python3
"""Config-3 SpMM suite: sum / mean / max forward, value and dense backward,
plus config 2 (100k x 100k, nnz 1M, F=64).  Kernel time by HIP events,
algorithmic bytes per SURVEY.md §8(d)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import algorithmic_bytes, event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)


def report(name, ms, nnz, alg):
    print(f"{name:34s} {ms:8.3f} ms  {nnz / ms / 1e6:7.2f} GEdges/s  {alg / ms / 1e9:6.2f} TB/s algorithmic "
          f"({alg / ms / 1e9 / 8 * 100:5.1f}% of 8 TB/s)")


# ---- config 2 ---------------------------------------------------------------
M, nnz, F = 100_000, 1_000_000, 64
rowptr, col, val = make_workload(M, M, nnz, F, 1, dev)
B = torch.randn(M, F, device=dev)
ops.spmm_sum(rowptr, col, val, B)
report("C2 spmm_sum 100k nnz=1M F=64", event_ms(lambda: ops.spmm_sum(rowptr, col, val, B), 200), nnz,
       algorithmic_bytes(nnz, M, F))

# ---- config 3 ---------------------------------------------------------------
M, nnz, F = 2_000_000, 20_000_000, 128
rowptr, col, val = make_workload(M, M, nnz, F, 2, dev)
B = torch.randn(M, F, device=dev)
G = torch.randn(M, F, device=dev)
for op, arg in (("sum", False), ("mean", False), ("max", True), ("min", True)):
    fn = getattr(ops, f"spmm_{op}")
    fn(rowptr, col, val, B)
    report(f"C3 spmm_{op} fwd", event_ms(lambda: fn(rowptr, col, val, B), 20), nnz,
           algorithmic_bytes(nnz, M, F, True, arg))
report("C3 spmm_sum fwd (value=None)", event_ms(lambda: ops.spmm_sum(rowptr, col, None, B), 20), nnz,
       algorithmic_bytes(nnz, M, F, False))

# backward pieces
ops.spmm_value_bw(None, rowptr, col, B, G, "sum")
report("C3 grad_value (spmm_value_bw)", event_ms(lambda: ops.spmm_value_bw(None, rowptr, col, B, G, "sum"), 20),
       nnz, nnz * (8 + 8 + 2 * 4 * F + 4))
row = ops.ptr2ind(rowptr, nnz)
a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
st = a.storage
t_csc = event_ms(lambda: (st.clear_cache_(), st.csr2csc(), st.colptr(), st._row_in_csc_order()), 3)
print(f"{'C3 CSC build (csr2csc+colptr+row_csc)':34s} {t_csc:8.3f} ms  (one-off per matrix, cached)")
row_csc, colptr, perm = st._row_in_csc_order(), st.colptr(), st.csr2csc()


def grad_mat():
    w = ops.transpose_weights(val, perm, row_csc, rowptr, False)
    return ops.spmm_sum(colptr, row_csc, w, G)


grad_mat()
report("C3 grad_mat (A^T gOut over CSC)", event_ms(grad_mat, 20), nnz,
       algorithmic_bytes(nnz, M, F) + nnz * (8 + 4 + 4))
out, arg = ops.spmm_max(rowptr, col, val, B)
ops.spmm_minmax_bw(col, val, B, G, arg)
print(f"{'C3 spmm_max bwd (float atomics)':34s} {event_ms(lambda: ops.spmm_minmax_bw(col, val, B, G, arg), 5):8.3f} ms")
tags = st._csc_edge_tags()
inv = st.csc2csr()
ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, perm, tags, val, B, G, arg, csc2csr=inv)
t_csc_bw = event_ms(lambda: ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, perm, tags, val, B, G, arg, csc2csr=inv), 5)
print(f"{'C3 spmm_max bwd (one CSC pass)':34s} {t_csc_bw:8.3f} ms  (production: no atomics, reproducible)")

# end-to-end autograd step (fwd + bwd of sum)
v = val.clone().requires_grad_()
Bt = B.clone().requires_grad_()
a2 = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
a2.storage._csr2csc, a2.storage._colptr, a2.storage._row_csc, a2.storage._csc2csr = perm, colptr, row_csc, inv


def fwd_bwd(reduce="sum"):
    v.grad = Bt.grad = None
    a2.matmul(Bt, reduce).backward(G)


for red in ("sum", "max"):
    fwd_bwd(red)
    ms = event_ms(lambda: fwd_bwd(red), 10)
    print(f"{'C3 spmm_' + red + ' fwd+bwd (autograd)':34s} {ms:8.3f} ms  {nnz / ms / 1e6:7.2f} GEdges/s")

# fixed adjacency (values not trained): grad wrt the dense operand only
a3 = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
a3.storage._csr2csc, a3.storage._colptr, a3.storage._row_csc = perm, colptr, row_csc


def fwd_bwd_fixed():
    Bt.grad = None
    (a3 @ Bt).backward(G)


fwd_bwd_fixed()
ms = event_ms(fwd_bwd_fixed, 10)
print(f"{'C3 spmm_sum fwd+bwd, fixed A':34s} {ms:8.3f} ms  {nnz / ms / 1e6:7.2f} GEdges/s")
