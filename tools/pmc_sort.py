#!/usr/bin/env python3
"""Workload for --pmc passes on the sort: 3 index_sort calls, 100M 48-bit keys."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from paddle_sparse_amd import ops  # noqa: E402

n = 100_000_000
g = torch.Generator(device="cuda").manual_seed(0)
keys = torch.randint(0, 1 << 48, (n,), generator=g, device="cuda")
for _ in range(3):
    ops.index_sort(keys, 1 << 48, with_sorted_inputs=True)
torch.cuda.synchronize()
print("done")
