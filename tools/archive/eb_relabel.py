#!/usr/bin/env python3
"""Is R-MAT slow because of its degree distribution or because of WHERE its hot
columns sit in memory?  Same graph with (a) the column ids relabelled by a random
permutation (same degree multisets, hot rows of B spread over the address space),
(b) columns replaced by uniform random ones (row degrees kept)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import algorithmic_bytes, event_ms  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402
from eb_probe import rmat  # noqa: E402

dev = torch.device("cuda", 0)
F = 128
N, row, col, val = rmat(21, 20_000_000)
rowptr = ops.ind2ptr(row, N)
nnz = col.numel()
g = torch.Generator(device=dev).manual_seed(7)
perm = torch.randperm(N, generator=g, device=dev)
cases = [("R-MAT as generated", col), ("columns relabelled randomly", perm[col].contiguous()),
         ("uniform random columns", torch.randint(0, N, (nnz,), generator=g, device=dev))]
B = torch.randn(N, F, device=dev)
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 30]
for name, c in cases:
    for variant in variants:
        ops.spmm_set_variant(variant)
        ops.spmm_sum(rowptr, c, val, B, row=row)
        ms = event_ms(lambda: ops.spmm_sum(rowptr, c, val, B, row=row), 20)
        print(f"{name:30s} variant {variant:3d}: {ms:.3f} ms  {algorithmic_bytes(nnz, N, F) / ms / 1e9:.2f} TB/s algorithmic", flush=True)
ops.spmm_set_variant(0)
