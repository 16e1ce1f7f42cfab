#!/usr/bin/env python3
"""BASELINE config 2 (spmm_sum, CSR 100k x 100k, nnz 1 M, F = 64, forward): the raw op (long-row scratch always
brought) and the tensor surface (which knows the longest row and skips the long-row launches), us per call."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)
M = N = 100_000
nnz, F = 1_000_000, 64
rowptr, col, val = make_workload(M, N, nnz, F, 1, dev)
B = torch.randn(N, F, device=dev)
a = SparseTensor(rowptr=rowptr, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
nb = nnz * (12 + 4 * F) + M * (8 + 4 * F)
for name, fn in (("ops.spmm_sum (raw op)", lambda: ops.spmm_sum(rowptr, col, val, B)),
                 ("SparseTensor.matmul under no_grad", lambda: a.matmul(B))):
    with torch.no_grad():
        for _ in range(20):
            fn()
        ms = event_ms(fn, 200)
    print(f"{name:40s} {ms * 1e3:7.1f} us  {nnz / ms / 1e6:6.2f} GEdges/s  {nb / ms / 1e6 / 8000:.3f} of 8 TB/s")
assert torch.equal(ops.spmm_sum(rowptr, col, val, B), a.matmul(B))
