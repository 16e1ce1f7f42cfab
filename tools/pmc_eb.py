#!/usr/bin/env python3
"""One SpMM workload for a rocprofv3 pass: tools/pmc_eb.py <c3|rmat> <variant> [op] [launches]."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402
from eb_probe import rmat  # noqa: E402

which, variant = sys.argv[1], int(sys.argv[2])
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
torch.cuda.synchronize()
for _ in range(launches):
    out = fn(rowptr, col, val, B, row=row)
torch.cuda.synchronize()
print(which, variant, op, "done")
