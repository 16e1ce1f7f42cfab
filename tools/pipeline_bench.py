#!/usr/bin/env python3
"""BASELINE config 5: coalesce + transpose + ind2ptr on a 100M-edge power-law
(R-MAT) COO with fp32 values, one MI355X.

pipeline = transpose(index, value, N, N)   (swap, sort by (row, col), add duplicates)
           -> ind2ptr(row', N)
Reports MEdges/s per stage and for the whole pipeline, the fraction of the
HBM roofline on the floor model (read inputs + write outputs once) and on the
LSD-radix model, next to a torch (rocPRIM) formulation of the
same pipeline as a GPU yardstick.
"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from paddle_sparse_amd import ops, transpose  # noqa: E402


def rmat(scale: int, nedges: int, seed: int, device, a=0.57, b=0.19, c=0.19):
    """R-MAT (a, b, c, d) edge generator, unsorted, natural duplicates."""
    g = torch.Generator(device=device).manual_seed(seed)
    row = torch.zeros(nedges, dtype=torch.int64, device=device)
    col = torch.zeros(nedges, dtype=torch.int64, device=device)
    for bit in range(scale):
        r = torch.rand(nedges, generator=g, device=device)
        right = ((r >= a) & (r < a + b)) | (r >= a + b + c)   # quadrants b, d
        down = r >= a + b                                     # quadrants c, d
        row |= down.to(torch.int64) << bit
        col |= right.to(torch.int64) << bit
    return row, col


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--edges", type=int, default=100_000_000)
args = ap.parse_args()
dev = torch.device("cuda", 0)
N = 1 << args.scale
n = args.edges
row, col = rmat(args.scale, n, 4, dev)
val = torch.randn(n, generator=torch.Generator(device=dev).manual_seed(5), device=dev)
index = torch.stack([row, col])
del row, col
torch.cuda.synchronize()


def ours():
    idx, v = transpose(index, val, N, N)
    rowptr = ops.ind2ptr(idx[0].contiguous(), N)
    return idx, v, rowptr


def torch_yardstick():
    key = index[1] * N + index[0]
    skey, perm = torch.sort(key, stable=True)
    uniq, inv = torch.unique_consecutive(skey, return_inverse=True)
    v = torch.zeros(uniq.numel(), device=dev).index_add_(0, inv, val[perm])
    r = torch.div(uniq, N, rounding_mode="floor")
    c = uniq - r * N
    rowptr = torch.searchsorted(r, torch.arange(N + 1, device=dev))
    return torch.stack([r, c]), v, rowptr


t_ours, (idx, v, rowptr) = timed(ours)
# sum of |terms| per output entry: what the rounding of either summation order is relative to
_key = index[1] * N + index[0]  # the transposed entry of (row, col)
_uniq, _inv = torch.unique(_key, return_inverse=True)
abs_sum = torch.zeros(_uniq.numel(), device=dev).index_add_(0, _inv, val.abs())
del _key, _uniq, _inv
t_torch, (idx_t, v_t, rowptr_t) = timed(torch_yardstick)
nnz2 = idx.shape[1]
print(f"R-MAT scale {args.scale}: {n} edges -> {nnz2} after coalesce ({100 * (1 - nnz2 / n):.1f}% duplicates)")
print("index match torch:", bool(torch.equal(idx, idx_t)), " rowptr match:", bool(torch.equal(rowptr, rowptr_t)),
      " value max err / sum|terms|:", float(((v - v_t).abs() / (abs_sum + 1e-30)).max()),
      "(the bar: 1e-5; relative to |result| it reads", float(((v - v_t).abs() / (v_t.abs() + 1e-6)).max()),
      "where terms cancel)")

# stage timings of our pipeline
r_in, c_in = index[1].contiguous(), index[0].contiguous()
t_keys, (keys, flag) = timed(lambda: ops.make_keys(r_in, c_in, N, check_sorted=True))
t_sort, (skeys, perm) = timed(lambda: ops.index_sort(keys, N * N, with_sorted_inputs=True))
t_uniq, (cnt, ptr, r2, c2) = timed(lambda: ops.unique_sorted(skeys, N))
t_seg, _ = timed(lambda: ops.segment_csr(val, ptr, "add", perm=perm))
t_i2p, _ = timed(lambda: ops.ind2ptr(r2, N))
t_pairs, (_, sval) = timed(lambda: ops.sort_pairs(keys, val, N * N))
t_seg2, _ = timed(lambda: ops.segment_csr(sval, ptr, "add"))
print(f"  (production path for fp32 scalar values: sort_pairs {t_pairs:.3f} ms + segment_csr {t_seg2:.3f} ms; "
      f"the two lines index_sort / segment_csr(perm) below are the generic-dtype path)")
passes = (2 * args.scale + 7) // 8
floor_bytes = n * (8 + 8 + 4) + nnz2 * (8 + 8 + 4) + nnz2 * 8 + (N + 1) * 8
radix_bytes = n * 24 + n * (32 * passes - 4) + n * 16 + n * (8 + 4 + 64) + nnz2 * 28 + nnz2 * 8 + (N + 1) * 8
for name, t in (("make_keys", t_keys), ("index_sort", t_sort), ("unique_sorted", t_uniq),
                ("segment_csr(perm)", t_seg), ("ind2ptr", t_i2p)):
    print(f"  {name:18s} {t:8.3f} ms  {n / t / 1e3:9.1f} MEdges/s")
print(f"pipeline (ours) : {t_ours:8.3f} ms  {n / t_ours / 1e3:9.1f} MEdges/s  "
      f"floor-model {floor_bytes / t_ours / 1e9 / 8 * 100:.1f}% of 8 TB/s, radix-model {radix_bytes / t_ours / 1e9 / 8 * 100:.1f}%")
print(f"pipeline (torch): {t_torch:8.3f} ms  {n / t_torch / 1e3:9.1f} MEdges/s")
