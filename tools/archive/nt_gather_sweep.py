#!/usr/bin/env python3
"""Where do non-temporal gathers of the dense rows start to pay?  The rank-local
SpMM of bench.py's weak-scaling shapes (2 M rows, 20 M edges, F = 128; B has
world x 2 M rows) plus C4's shard (F = 256), ordinary against non-temporal
gathers (variant 18), one process, interleaved."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from paddle_sparse_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, NNZ = 2_000_000, 20_000_000


def ms(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


for N, F in ((2_000_000, 128), (3_000_000, 128), (4_000_000, 128), (6_000_000, 128), (8_000_000, 128), (16_000_000, 128),
             (2_000_000, 256), (4_000_000, 256), (16_000_000, 256), (8_000_000, 64), (16_000_000, 64)):
    g = torch.Generator(device=dev).manual_seed(3)
    row = torch.randint(0, M, (NNZ,), generator=g, device=dev).sort().values
    col = torch.randint(0, N, (NNZ,), generator=g, device=dev)
    val = torch.randn(NNZ, generator=g, device=dev)
    rowptr = ops.ind2ptr(row, M)
    B = torch.randn(N, F, generator=g, device=dev)
    t = {}
    for rnd in range(2):
        for v in (0, 18):
            ops.spmm_set_variant(v)
            t.setdefault(v, []).append(ms(lambda: ops.spmm_sum(rowptr, col, val, B)))
    ops.spmm_set_variant(0)
    a, b = min(t[0]), min(t[18])
    print(f"B = {N} x {F} fp32 = {N * F * 4 / 2**30:6.2f} GiB: ordinary gathers {a:.3f} ms, non-temporal {b:.3f} ms  ({a / b:.3f}x)",
          flush=True)
    del B, row, col, val, rowptr
    torch.cuda.empty_cache()
