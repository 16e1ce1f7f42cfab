#!/usr/bin/env python3
"""index_sort throughput vs torch.sort (rocPRIM onesweep) as a yardstick."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from paddle_sparse_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 48
g = torch.Generator(device="cuda").manual_seed(0)
keys = torch.randint(0, 1 << bits, (n,), generator=g, device="cuda")


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_ours = timeit(lambda: ops.index_sort(keys, 1 << bits, with_sorted_inputs=True))
t_ours_perm = timeit(lambda: ops.index_sort(keys, 1 << bits))
t_torch = timeit(lambda: torch.sort(keys, stable=True))
passes = (bits + 7) // 8
model = n * (32 * passes - 4)  # 8n hist + 12n read + 12n write per pass (pass 0 reads no idx)
print(f"n={n} bits={bits} passes={passes}")
print(f"index_sort (sorted+perm): {t_ours:.3f} ms  {n / t_ours / 1e6:.2f} GKeys/s  {model / t_ours / 1e9:.2f} TB/s (radix model)")
print(f"index_sort (perm only)  : {t_ours_perm:.3f} ms")
print(f"torch.sort stable (rocPRIM, 64-bit full key): {t_torch:.3f} ms  {n / t_torch / 1e6:.2f} GKeys/s")
srt, perm = ops.index_sort(keys, 1 << bits, with_sorted_inputs=True)
ts, tp = torch.sort(keys, stable=True)
print("match torch.sort:", bool(torch.equal(srt, ts)), bool(torch.equal(perm, tp)))
