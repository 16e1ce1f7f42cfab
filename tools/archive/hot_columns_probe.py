#!/usr/bin/env python3
"""Would a compact copy of the HOT rows of B cure the R-MAT slowdown?  The hub columns of
R-MAT as generated sit at ids with few bits set, i.e. at addresses that fall on few memory
channels.  Emulation: the k most referenced columns are redirected to copies of their rows
appended contiguously behind B (col' = N + slot); everything else unchanged."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402
from eb_probe import rmat  # noqa: E402

dev = torch.device("cuda", 0)
F = 128
N, row, col, val = rmat(21, 20_000_000)
rowptr = ops.ind2ptr(row, N)
nnz = col.numel()
B = torch.randn(N, F, device=dev)
count = ops.bincount(col, N)
order = torch.argsort(count, descending=True)
ref = ops.spmm_sum(rowptr, col, val, B, row=row, algo="edge_ranges")
for algo in ("edge_ranges", "row_waves"):
    base = event_ms(lambda: ops.spmm_sum(rowptr, col, val, B, row=row, algo=algo), 20)
    print(f"[{algo}] as generated: {base:.3f} ms", flush=True)
    for k in (1024, 8192, 65536, 262144):
        hot = order[:k]
        share = float(count[hot].sum()) / nnz
        slot = torch.full((N,), -1, dtype=torch.int64, device=dev)
        slot[hot] = torch.arange(k, device=dev)
        s = slot[col]
        col2 = torch.where(s >= 0, N + s, col).contiguous()
        B2 = torch.cat([B, B[hot]]).contiguous()
        out = ops.spmm_sum(rowptr, col2, val, B2, row=row, algo=algo)
        assert torch.equal(out, ref) or algo != "edge_ranges"
        ms = event_ms(lambda: ops.spmm_sum(rowptr, col2, val, B2, row=row, algo=algo), 20)
        pack = event_ms(lambda: ops.gather_rows(B, hot), 20)
        print(f"[{algo}] {k:7d} hottest columns ({share * 100:4.1f} % of the entries) served from a compact copy: "
              f"{ms:.3f} ms (+ {pack * 1e3:.0f} us to pack the copy)", flush=True)
