#!/usr/bin/env python3
"""Tile shape of the single-sweep radix pass on mid-size inputs: index_sort of n keys
(3 passes) with every shape the pass kernel is built for, back to back on one stream."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import _lib, ops  # noqa: E402

SHAPES = {0: "512x16", 5: "1024x8", 8: "512x4", 9: "512x8", 10: "256x8", 11: "1024x4"}
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for n in (20_000, 50_000, 100_000, 262_144, 524_287, 1_000_000):
    bound = max(10 * n, 1 << 17)
    keys = torch.randint(0, bound, (n,), generator=g, device="cuda")
    ref = None
    line = f"n {n:>9,d}:"
    for variant, name in SHAPES.items():
        prev = lib.psa_sort_set_variant(variant)
        try:
            for _ in range(5):
                _, perm = ops.index_sort(keys, bound)
            reps = 200 if n <= 1_000_000 else 50
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                _, perm = ops.index_sort(keys, bound)
            b.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = perm
                assert torch.equal(keys[perm], torch.sort(keys, stable=True).values)
            ok = torch.equal(perm, ref)
            line += f"  {name} {a.elapsed_time(b) / reps * 1e3:7.1f} us{'' if ok else ' WRONG'}"
        finally:
            lib.psa_sort_set_variant(prev)
    print(line, flush=True)
