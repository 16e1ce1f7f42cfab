#!/usr/bin/env python3
"""index_sort / coalesce at 1e9 keys (8 GB of keys, 24 GB of sort workspace):
32-bit arithmetic on element offsets would break here (2^30 < n < 2^31, byte
offsets far past 2^32).  Checks order, permutation validity and stability."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(4)
keys = torch.randint(0, 1 << 40, (n,), generator=g, device="cuda")
keys[::1000] = 12345  # a million equal keys spread over the whole array: stability is observable
torch.cuda.synchronize()
t0 = time.perf_counter()
skeys, perm = ops.index_sort(keys, 1 << 40, with_sorted_inputs=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"n = {n}: index_sort, first call (allocates 40 GB of outputs and workspace) {dt * 1e3:.1f} ms")
del skeys, perm
torch.cuda.synchronize()
t0 = time.perf_counter()
skeys, perm = ops.index_sort(keys, 1 << 40, with_sorted_inputs=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"n = {n}: index_sort {dt * 1e3:.1f} ms = {n / dt / 1e9:.2f} GKeys/s (5 passes of 8 bits)")
print("sorted:", bool((skeys[1:] >= skeys[:-1]).all()), " keys[perm] == sorted:", bool(torch.equal(keys[perm], skeys)))
eq = skeys[1:] == skeys[:-1]
print("stable (perm increasing inside runs of equal keys):", bool((perm[1:][eq] > perm[:-1][eq]).all()),
      f" ({int(eq.sum())} adjacent equal pairs)")
chk = torch.zeros(n, dtype=torch.int8, device="cuda")
chk[perm] = 1
print("perm is a permutation:", bool(chk.all()), " peak HBM %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
