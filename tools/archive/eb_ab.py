#!/usr/bin/env python3
"""A/B timings of the edge-balanced SpMM variants (uniform config 3 and R-MAT 21)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import algorithmic_bytes, event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402
from eb_probe import rmat  # noqa: E402

dev = torch.device("cuda", 0)
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 30]
ops_list = sys.argv[2].split(",") if len(sys.argv) > 2 else ["spmm_sum", "spmm_max"]
F = 128
graphs = []
M, nnz = 2_000_000, 20_000_000
rowptr, col, val = make_workload(M, M, nnz, F, 2, dev)
graphs.append(("uniform C3", M, rowptr, ops.ptr2ind(rowptr, nnz), col, val))
N, row, col2, val2 = rmat(21, 20_000_000)
graphs.append(("R-MAT 21", N, ops.ind2ptr(row, N), row, col2, val2))
for name, M, rowptr, row, col, val in graphs:
    nnz = col.numel()
    B = torch.randn(M, F, device=dev)
    for rep in range(2):
        for variant in variants:
            ops.spmm_set_variant(variant)
            for op in ops_list:
                fn = getattr(ops, op)
                fn(rowptr, col, val, B, row=row)
                ms = event_ms(lambda: fn(rowptr, col, val, B, row=row), 20)
                alg = algorithmic_bytes(nnz, M, F, True, op == "spmm_max")
                print(f"{name:11s} variant {variant:3d} {op}: {ms:.3f} ms  {nnz / ms / 1e6:.2f} GEdges/s  "
                      f"{alg / ms / 1e9:.2f} TB/s algorithmic", flush=True)
ops.spmm_set_variant(0)
