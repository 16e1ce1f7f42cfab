#!/usr/bin/env python3
"""SpMM on a power-law (R-MAT) graph: how much do long rows cost the
one-wave-per-row kernel?  2M nodes (scale 21), 20M edges, F=128."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import algorithmic_bytes, event_ms  # noqa: E402
from paddle_sparse_amd import coalesce, ops  # noqa: E402

dev = torch.device("cuda", 0)
scale, n, F = 21, 20_000_000, 128
N = 1 << scale
g = torch.Generator(device=dev).manual_seed(4)
row = torch.zeros(n, dtype=torch.int64, device=dev)
col = torch.zeros(n, dtype=torch.int64, device=dev)
for bit in range(scale):
    r = torch.rand(n, generator=g, device=dev)
    right = ((r >= 0.57) & (r < 0.76)) | (r >= 0.95)
    down = r >= 0.76
    row |= down.to(torch.int64) << bit
    col |= right.to(torch.int64) << bit
index, val = coalesce(torch.stack([row, col]), torch.randn(n, generator=g, device=dev), N, N)
nnz = index.shape[1]
rowptr = ops.ind2ptr(index[0].contiguous(), N)
deg = rowptr[1:] - rowptr[:-1]
print(f"R-MAT scale {scale}: nnz={nnz}, max deg={int(deg.max())}, rows with deg>4096: {int((deg > 4096).sum())}, "
      f"empty rows: {int((deg == 0).sum())}")
B = torch.randn(N, F, device=dev)
c = index[1].contiguous()
for variant, label in ((10, "one wave per row, any length"), (15, "long rows chunked, separate launches"),
                       (0, "production: long rows chunked, roles fused"),
                       (13, "2 rows per wave"), (11, "4 rows per wave"), (12, "8 rows per wave"),
                       (14, "fused roles: chunks + rows in one launch"),
                       (20, "fused roles, 512 chunk workgroups"), (21, "fused roles, 1024 chunk workgroups"),
                       (22, "fused roles, 1536 chunk workgroups"),
                       (7, "two rows per wave side by side (multirow kernel at K = 128)")):
    ops.spmm_set_variant(variant)
    for op in ("spmm_sum", "spmm_max"):
        fn = getattr(ops, op)
        fn(rowptr, c, val, B)
        ms = event_ms(lambda: fn(rowptr, c, val, B), 20)
        alg = algorithmic_bytes(nnz, N, F, True, op == "spmm_max")
        print(f"[{label}] {op}: {ms:.3f} ms  {nnz / ms / 1e6:.2f} GEdges/s  {alg / ms / 1e9:.2f} TB/s algorithmic")
ops.spmm_set_variant(0)
