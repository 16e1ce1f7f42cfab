#!/usr/bin/env python3
"""One SpMM workload for a rocprofv3 pass: tools/archive/pmc_eb.py <c3|rmat|rmat_relabel> <variant|hot> [op] [launches].
variant: a psa_spmm_set_variant id (0 = row waves, 30 = edge ranges); "hot" = edge ranges reading the hub
columns from the compact copy (plan of SparseStorage._hot_columns, packed inside every launch)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402
from eb_probe import rmat  # noqa: E402

which, hot = sys.argv[1], sys.argv[2] == "hot"
variant = 0 if hot else int(sys.argv[2])
op = sys.argv[3] if len(sys.argv) > 3 else "spmm_sum"
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda", 0)
F = 128
if which in ("rmat", "rmat_relabel"):
    M, row, col, val = rmat(21, 20_000_000)
    rowptr = ops.ind2ptr(row, M)
    if which == "rmat_relabel":  # hot columns spread over the address space (Graph500-style vertex relabelling, columns only)
        g = torch.Generator(device=dev).manual_seed(7)
        col = torch.randperm(M, generator=g, device=dev)[col].contiguous()
else:
    M = 2_000_000
    rowptr, col, val = make_workload(M, M, 20_000_000, F, 2, dev)
    row = ops.ptr2ind(rowptr, col.numel())
B = torch.randn(M, F, device=dev)
ops.spmm_set_variant(variant)
fn = getattr(ops, op)
if hot:
    from paddle_sparse_amd import SparseTensor

    plan = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(M, M), is_sorted=True,
                        trust_data=True).storage._hot_columns()
    hot_cols, col_eff = plan
    fn = lambda rp, c, v, b, row=None: getattr(ops, op)(rp, col_eff, v, b, row=row, algo="edge_ranges") \
        if False else ops._spmm(op.split("_")[1], rp, col_eff, v, b, row=row, algo="edge_ranges",
                                hot_rows=ops.gather_rows(b, hot_cols))  # noqa: E731
torch.cuda.synchronize()
for _ in range(launches):
    out = fn(rowptr, col, val, B, row=row)
torch.cuda.synchronize()
print(which, variant, op, "done")
