#!/usr/bin/env python3
"""index_select / masked_select / narrow on the config-3 graph (2M x 2M, 20M
entries, fp32 values), against the same selections written with torch ops on the
COO arrays (the reference's formulation: boolean masks and repeat_interleave)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import SparseTensor  # noqa: E402

M, NNZ = 2_000_000, 20_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
key = torch.unique(torch.randint(0, M * M, (NNZ,), generator=g, device=dev))
row, col = torch.div(key, M, rounding_mode="floor"), key % M
val = torch.randn(row.numel(), generator=g, device=dev)
A = SparseTensor(row=row, col=col, value=val, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
A.storage.rowptr()
A.storage.colptr()
A.storage.csr2csc()


def ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


rows = torch.randperm(M, generator=g, device=dev)[:500_000]
mask = torch.rand(M, generator=g, device=dev) < 0.25
print(f"index_select(0, 500k random rows): {ms(lambda: A.index_select(0, rows)):.3f} ms")
print(f"index_select(1, 500k random cols): {ms(lambda: A.index_select(1, rows)):.3f} ms")
print(f"masked_select(0, 25% of rows):     {ms(lambda: A.masked_select(0, mask)):.3f} ms")
print(f"masked_select(1, 25% of cols):     {ms(lambda: A.masked_select(1, mask)):.3f} ms")
print(f"narrow(0, 500000, 500000):         {ms(lambda: A.narrow(0, 500_000, 500_000)):.3f} ms")
print(f"narrow(1, 500000, 500000):         {ms(lambda: A.narrow(1, 500_000, 500_000)):.3f} ms")
print(f"permute(random perm):              {ms(lambda: A.permute(torch.randperm(M, generator=g, device=dev)), 3):.3f} ms")
# correctness spot check against dense indexing on a corner
sub = A.index_select(0, rows[:50]).index_select(1, rows[50:120]).to_dense()
ref = torch.zeros(50, 70, device=dev)
r_map = {int(r): i for i, r in enumerate(rows[:50].tolist())}
c_map = {int(c): i for i, c in enumerate(rows[50:120].tolist())}
sel = torch.isin(row, rows[:50]) & torch.isin(col, rows[50:120])
for r, c, v in zip(row[sel].tolist(), col[sel].tolist(), val[sel].tolist()):
    ref[r_map[r], c_map[c]] = v
print("spot check vs dense indexing:", bool(torch.equal(sub, ref)))
