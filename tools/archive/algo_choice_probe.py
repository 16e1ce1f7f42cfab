#!/usr/bin/env python3
"""Does the per-matrix choice (SparseStorage._spmm_algo / _hot_columns) pick the faster forward on
graphs between "uniform" and "R-MAT"?  Zipf row degrees and Zipf column popularity at several
exponents and average degrees (2 M rows, ~20 M entries, F = 128): both kernel families, the
tensor surface, and what the row statistics chose."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms  # noqa: E402
from paddle_sparse_amd import SparseTensor, coalesce, ops  # noqa: E402

dev = torch.device("cuda", 0)
N, F, TARGET = 2_000_000, 128, 20_000_000
g = torch.Generator(device=dev).manual_seed(11)


def zipf_ids(n, size, alpha):
    """n draws from {0..size-1} with P(i) ~ (i + 1)^-alpha, ids shuffled (hubs anywhere)."""
    if alpha == 0:
        return torch.randint(0, size, (n,), generator=g, device=dev)
    u = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    if abs(alpha - 1.0) < 1e-9:
        x = torch.exp(u * torch.log(torch.tensor(float(size), dtype=torch.float64, device=dev)))
    else:
        a = 1.0 - alpha
        x = ((size ** a - 1.0) * u + 1.0) ** (1.0 / a)
    ids = (x.long() - 1).clamp_(0, size - 1)
    return torch.randperm(size, generator=g, device=dev)[ids]


for row_alpha, col_alpha in ((0, 0), (0.5, 0.5), (0.8, 0.8), (1.0, 1.0), (1.2, 1.2), (0, 1.0), (1.0, 0)):
    row, col = zipf_ids(TARGET, N, row_alpha), zipf_ids(TARGET, N, col_alpha)
    index, val = coalesce(torch.stack([row, col]), torch.randn(TARGET, generator=g, device=dev), N, N)
    row, col = index[0].contiguous(), index[1].contiguous()
    rowptr = ops.ind2ptr(row, N)
    B = torch.randn(N, F, device=dev)
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
    empty, tiny, big, longest = ops.csr_row_stats(rowptr)
    line = (f"zipf rows {row_alpha} cols {col_alpha}: nnz {col.numel() / 1e6:.1f} M, empty {empty / N:.0%}, 1-2 {tiny / N:.0%}, "
            f">128 {big}, longest {longest}:")
    for algo in ("row_waves", "edge_ranges"):
        ops.spmm_sum(rowptr, col, val, B, row=row, algo=algo)
        line += f"  {algo} {event_ms(lambda: ops.spmm_sum(rowptr, col, val, B, row=row, algo=algo), 10):.3f} ms"
    with torch.no_grad():
        a.matmul(B, "sum")
        line += f"  surface {event_ms(lambda: a.matmul(B, 'sum'), 10):.3f} ms"
    line += f"  chose {a.storage._spmm_algo()}{' + hub copy' if a.storage._hot_columns() is not None else ''}"
    print(line, flush=True)
    del a, B, row, col, val, index, rowptr
