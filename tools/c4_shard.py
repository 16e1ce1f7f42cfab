#!/usr/bin/env python3
"""BASELINE config 4, one rank's share on one MI355X: the row block a GPU owns
of a 16M x 16M matrix with 160M edges split over 8 GPUs (2M rows, 20M edges,
columns over all 16M nodes) times the FULL dense B [16M, 256] fp32 (16.4 GB —
what the all-gather leaves in every GPU's HBM).  Prints the local SpMM time,
the algorithmic bytes of SURVEY.md §8(d) (22.78 GB per GPU) and the xGMI
arithmetic of the exchange that precedes it on the real 8-GPU job; rocSPARSE
(torch.sparse.mm on the same CSR) is timed beside it as a GPU yardstick and
as the correctness check at this size.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import ops  # noqa: E402

N, M_LOCAL, NNZ, F = 16_000_000, 2_000_000, 20_000_000, 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
row = torch.randint(0, M_LOCAL, (NNZ,), generator=g, device=dev).sort().values
col = torch.randint(0, N, (NNZ,), generator=g, device=dev)
val = torch.randn(NNZ, generator=g, device=dev)
rowptr = ops.ind2ptr(row, M_LOCAL)
B = torch.randn(N, F, generator=g, device=dev)


def ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


t = ms(lambda: ops.spmm_sum(rowptr, col, val, B))
alg = NNZ * (8 + 4 + 4 * F) + M_LOCAL * (8 + 4 * F)
print(f"C4 shard: {M_LOCAL} rows x {N} cols, nnz {NNZ}, F {F}: spmm_sum {t:.3f} ms = {NNZ / t / 1e6:.2f} GEdges/s, "
      f"{alg / 1e9:.2f} GB algorithmic -> {alg / t / 1e9:.2f} TB/s = {100 * alg / t / 1e9 / 8:.1f}% of 8 TB/s")
# the production path gathers an operand this far beyond the Infinity Cache non-temporally (spmm.hip, kNtGatherBytes)
for variant, what in ((17, "ordinary output stores and ordinary gathers"), (25, "one 64-lane K tile instead of two 32-lane tiles")):
    ops.spmm_set_variant(variant)
    tv = ms(lambda: ops.spmm_sum(rowptr, col, val, B))
    print(f"  variant {variant} ({what}): {tv:.3f} ms")
ops.spmm_set_variant(0)
out = ops.spmm_sum(rowptr, col, val, B)
csr = torch.sparse_csr_tensor(rowptr, col, val, size=(M_LOCAL, N))
t_ref = ms(lambda: torch.sparse.mm(csr, B), reps=3)
ref = torch.sparse.mm(csr, B)
scale = torch.sparse.mm(torch.sparse_csr_tensor(rowptr, col, val.abs(), size=(M_LOCAL, N)), B.abs())
err = ((out - ref).abs() / (scale + 1e-30)).max().item()
print(f"rocSPARSE (torch.sparse.mm, same CSR): {t_ref:.3f} ms; max |ours - rocSPARSE| / sum|terms| = {err:.2e}")
shard = N // 8 * F * 4
print(f"exchange on the 8-GPU job: each rank receives 7 x {shard / 1e9:.2f} GB of B per step; at the full 7 x 153 GB/s xGMI "
      f"ingest that is >= {7 * shard / (7 * 153e9) * 1e3:.1f} ms against {t:.2f} ms of local SpMM")
