#!/usr/bin/env python3
"""index_sort throughput (scatter variants A/B in one process, interleaved)
vs torch.sort (rocPRIM onesweep) as a yardstick."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from paddle_sparse_amd import _lib, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 48
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
keys = torch.randint(0, 1 << bits, (n,), generator=g, device="cuda")
lib = _lib.load()


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


passes = (bits + 7) // 8
model = n * (32 * passes - 4)  # 8n hist + 12n read + 12n write per pass (pass 0 reads no idx)
print(f"n={n} bits={bits} passes={passes}")
ts, tp = torch.sort(keys, stable=True)
times = {v: [] for v in variants}
for _ in range(3):
    for v in variants:
        lib.psa_sort_set_variant(v)
        times[v].append(timeit(lambda: ops.index_sort(keys, 1 << bits, with_sorted_inputs=True)))
for v in variants:
    lib.psa_sort_set_variant(v)
    srt, perm = ops.index_sort(keys, 1 << bits, with_sorted_inputs=True)
    ok = bool(torch.equal(srt, ts)) and bool(torch.equal(perm, tp))
    t = sorted(times[v])[1]
    print(f"variant {v}: {t:.3f} ms  {n / t / 1e6:.2f} GKeys/s  {model / t / 1e9:.2f} TB/s (radix model)  exact={ok}")
lib.psa_sort_set_variant(0)
t_torch = timeit(lambda: torch.sort(keys, stable=True))
print(f"torch.sort stable (rocPRIM, 64-bit full key): {t_torch:.3f} ms  {n / t_torch / 1e6:.2f} GKeys/s")
for name, k in (("sorted input", ts), ("constant input", torch.zeros_like(keys))):
    t = timeit(lambda: ops.index_sort(k, 1 << bits))
    print(f"{name}: {t:.3f} ms")
