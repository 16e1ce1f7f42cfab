#!/usr/bin/env python3
"""coalesce(index, value, m, n, "add") across sizes: where the one-workgroup
path, the always-sort range and the streaming chain take over, and what an
entry costs in each (uniform random COO, ~5 % duplicates, fp32 scalar values)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import paddle_sparse_amd as ps  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for nnz in (1_000, 10_000, 16_384, 16_385, 50_000, 100_000, 262_144, 262_145, 1_000_000, 4_000_000, 20_000_000, 100_000_000):
    m = n = max(int((10 * nnz) ** 0.5), 2)  # ~5 % of the entries collide
    index = torch.stack([torch.randint(0, m, (nnz,), generator=g, device="cuda"),
                         torch.randint(0, n, (nnz,), generator=g, device="cuda")])
    value = torch.randn(nnz, generator=g, device="cuda")
    small = nnz < 10_000_000
    for _ in range(100 if small else 3):  # steady state (the call reads its count back: it is synchronous)
        out = ps.coalesce(index, value, m, n)
    torch.cuda.synchronize()
    ts = []
    for _ in range(300 if small else 5):
        t0 = time.perf_counter()
        out = ps.coalesce(index, value, m, n)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print(f"nnz {nnz:>11,d} ({m} x {n}) -> {out[0].shape[1]:>11,d} entries: {t * 1e6:10.1f} us  "
          f"{t / nnz * 1e9:8.2f} ns/entry  {nnz / t / 1e9:7.3f} GEntries/s")
