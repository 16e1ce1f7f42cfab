#!/usr/bin/env python3
"""ptr2ind / ind2ptr on a uniform and on a power-law row pointer (R-MAT scale 21)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402
from eb_probe import rmat  # noqa: E402

dev = torch.device("cuda", 0)
rowptr, col, val = make_workload(2_000_000, 2_000_000, 20_000_000, 128, 2, dev)
N, row, col2, val2 = rmat(21, 20_000_000)
for name, rp, n in (("uniform", rowptr, col.numel()), ("R-MAT 21", ops.ind2ptr(row, N), row.numel())):
    r = ops.ptr2ind(rp, n)
    assert torch.equal(ops.ind2ptr(r, rp.numel() - 1), rp)
    if name != "uniform":
        assert torch.equal(r, row)
    print(f"{name}: ptr2ind {event_ms(lambda: ops.ptr2ind(rp, n), 50) * 1e3:.1f} us, "
          f"ind2ptr {event_ms(lambda: ops.ind2ptr(r, rp.numel() - 1), 50) * 1e3:.1f} us", flush=True)
