#!/usr/bin/env python3
"""Does giving each XCD a contiguous eighth of the rows (variant 16) pay?
spmm_sum, 2M x 2M, 20M edges, F = 128 on three column patterns: uniform random
(no locality), banded (col within +-w of row) and communities (blocks of nodes
with mostly internal edges)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import ops  # noqa: E402

M, NNZ, F = 2_000_000, 20_000_000, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
row = torch.randint(0, M, (NNZ,), generator=g, device=dev).sort().values
rowptr = ops.ind2ptr(row, M)
val = torch.randn(NNZ, generator=g, device=dev)
B = torch.randn(M, F, generator=g, device=dev)


def ms(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def cols(kind):
    if kind == "uniform":
        return torch.randint(0, M, (NNZ,), generator=g, device=dev)
    if kind.startswith("banded"):
        w = int(kind.split("_")[1])
        return (row + torch.randint(-w, w + 1, (NNZ,), generator=g, device=dev)) % M
    size = int(kind.split("_")[1])  # communities of `size` nodes, 90 % internal edges
    inside = (row // size) * size + torch.randint(0, size, (NNZ,), generator=g, device=dev)
    anywhere = torch.randint(0, M, (NNZ,), generator=g, device=dev)
    return torch.where(torch.rand(NNZ, generator=g, device=dev) < 0.9, inside, anywhere) % M


for kind in ("uniform", "banded_200000", "banded_20000", "banded_2000", "community_65536", "community_4096"):
    col = cols(kind)
    res = {}
    for variant in (0, 16):
        ops.spmm_set_variant(variant)
        res[variant] = ms(lambda: ops.spmm_sum(rowptr, col, val, B))
    ops.spmm_set_variant(0)
    a = ops.spmm_sum(rowptr, col, val, B)
    ops.spmm_set_variant(16)
    b = ops.spmm_sum(rowptr, col, val, B)
    ops.spmm_set_variant(0)
    print(f"{kind:18s} round-robin rows {res[0]:6.3f} ms | XCD-contiguous rows {res[16]:6.3f} ms "
          f"({res[0] / res[16]:4.2f}x)  identical: {bool(torch.equal(a, b))}")
