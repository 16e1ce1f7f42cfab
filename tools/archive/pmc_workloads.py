#!/usr/bin/env python3
"""Workloads for the rocprofv3 --pmc passes (run one per profiler process).

  calib : permutation-matrix SpMM (M = N = 2M, nnz = M, F = 128).  Every row
          of B is read exactly once and B (1 GB) exceeds the 256 MiB Infinity
          Cache, so the HBM bytes are KNOWN: reads = N*4F + nnz*12 + (M+1)*8,
          writes = M*4F.  Calibrates FETCH_SIZE / WRITE_SIZE for this access
          pattern (MI355X_MICROARCH.md §HBM says to).
  c3    : BASELINE config 3 (the bench workload), a few launches.
  rmat  : the R-MAT scale-21 graph of tools/archive/spmm_rmat.py.
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
op = sys.argv[2] if len(sys.argv) > 2 else "spmm_sum"
dev = torch.device("cuda", 0)
M = N = 2_000_000
F = 128
if which == "calib":
    g = torch.Generator(device=dev).manual_seed(0)
    rowptr = torch.arange(M + 1, device=dev)
    col = torch.randperm(N, generator=g, device=dev)
    val = torch.randn(M, generator=g, device=dev)
elif which == "rmat":  # the power-law graph of tools/archive/spmm_rmat.py (scale 21, 20 M edges before coalescing)
    from paddle_sparse_amd import coalesce

    M = N = 1 << 21
    g = torch.Generator(device=dev).manual_seed(4)
    n = 20_000_000
    row = torch.zeros(n, dtype=torch.int64, device=dev)
    col = torch.zeros(n, dtype=torch.int64, device=dev)
    for bit in range(21):
        r = torch.rand(n, generator=g, device=dev)
        row |= (r >= 0.76).to(torch.int64) << bit
        col |= (((r >= 0.57) & (r < 0.76)) | (r >= 0.95)).to(torch.int64) << bit
    index, val = coalesce(torch.stack([row, col]), torch.randn(n, generator=g, device=dev), N, N)
    rowptr, col = ops.ind2ptr(index[0].contiguous(), M), index[1].contiguous()
else:
    rowptr, col, val = make_workload(M, N, 20_000_000, F, 2, dev)
B = torch.randn(N, F, device=dev)
fn = getattr(ops, op)
torch.cuda.synchronize()
for _ in range(5):
    out = fn(rowptr, col, val, B)
torch.cuda.synchronize()
print(which, op, "done")
