#!/usr/bin/env python3
"""A/B the SpMM kernel variants in ONE process, interleaved rounds
(cdna_hip_programming.md §5.4 rule 24), on BASELINE config 3 (or --small)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import algorithmic_bytes, make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,2,3,4")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--M", type=int, default=2_000_000)
ap.add_argument("--nnz", type=int, default=20_000_000)
ap.add_argument("--F", type=int, default=128)
ap.add_argument("--op", default="spmm_sum")
args = ap.parse_args()

dev = torch.device("cuda", 0)
rowptr, col, val = make_workload(args.M, args.M, args.nnz, args.F, 2, dev)
B = torch.randn(args.M, args.F, device=dev)
fn = getattr(ops, args.op)
variants = [int(v) for v in args.variants.split(",")]
alg = algorithmic_bytes(args.nnz, args.M, args.F, True, args.op in ("spmm_max", "spmm_min"))

# copy bandwidth yardstick (read 1x + write 1x)
x = torch.empty(256 * 1024 * 1024, device=dev)
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    y.copy_(x)
b.record()
torch.cuda.synchronize()
print(f"copy 1GiB: {2 * x.numel() * 4 * 10 / (a.elapsed_time(b) * 1e-3) / 1e12:.2f} TB/s (r+w)")
del x, y

times = {v: [] for v in variants}
for r in range(args.rounds):
    for v in variants:
        ops.spmm_set_variant(v)
        fn(rowptr, col, val, B)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn(rowptr, col, val, B)
        b.record()
        torch.cuda.synchronize()
        times[v].append(a.elapsed_time(b) / args.iters)
for v in variants:
    t = sorted(times[v])
    med, mn = t[len(t) // 2], t[0]
    print(f"variant {v}: median {med:.3f} ms  min {mn:.3f} ms  "
          f"{args.nnz / med / 1e6:.2f} GEdges/s  {alg / med / 1e9:.2f} TB/s algorithmic "
          f"({alg / med / 1e9 / 8.0 * 100:.1f}% of 8 TB/s)")
