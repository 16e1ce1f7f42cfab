#!/usr/bin/env python3
"""spspmm (A @ A of a uniform random graph) on one MI355X: stage times,
products/s, and the row-by-row CPU oracle beside it (single thread, bounded
by --cpu-rows).

    python tests/perf/spspmm_bench.py --nodes 2000000 --edges 20000000
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402  (CPU baseline leg only)
from paddle_sparse_amd import ops, spspmm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=2_000_000)
ap.add_argument("--edges", type=int, default=20_000_000)
ap.add_argument("--cpu-rows", type=int, default=200_000)
args = ap.parse_args()
M, nnz = args.nodes, args.edges
rng = np.random.default_rng(6)
key = np.unique(rng.integers(0, M * M, nnz))
index_h = np.stack([key // M, key % M])
val_h = rng.standard_normal(key.size).astype(np.float32)
index, val = torch.from_numpy(index_h).cuda(), torch.from_numpy(val_h).cuda()
E = key.size


def gpu_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out


t_all, (idxC, valC) = gpu_ms(lambda: spspmm(index, val, index, val, M, M, M))
row, col = index[0].contiguous(), index[1].contiguous()
rowptr = ops.ind2ptr(row, M)
t_cnt, counts = gpu_ms(lambda: ops.spspmm_count(col, rowptr))
t_scan, offsets = gpu_ms(lambda: ops.count2ptr(counts))
total = int(offsets[-1])
t_own, owner = gpu_ms(lambda: ops.ptr2ind(offsets, total))
t_exp, (keys, vals) = gpu_ms(lambda: ops.spspmm_expand(row, col, val, rowptr, col, val, offsets, owner, total, M,
                                                       torch.float32))
t_sort, (skeys, svals) = gpu_ms(lambda: ops.sort_pairs(keys, vals, M * M))
t_uniq, (cnt, ptr, r2, c2) = gpu_ms(lambda: ops.unique_sorted(skeys, M))
t_seg, _ = gpu_ms(lambda: ops.segment_csr(svals, ptr, "sum"))
print(f"A @ A, A = {M} x {M} with {E} entries: {total} products -> {idxC.shape[1]} entries")
for name, t in (("count", t_cnt), ("offsets (scan)", t_scan), ("owner (ptr2ind)", t_own), ("expand", t_exp),
                ("sort_pairs", t_sort), ("unique", t_uniq), ("segment sum", t_seg)):
    print(f"  {name:18s} {t:8.3f} ms")
print("  (the row-order walk above sorts on row * n + col; the production path for 4-byte values walks the")
print("   CSC views and sorts on the row field of (row << 32 | col) only:)")
from paddle_sparse_amd import SparseTensor  # noqa: E402

A = SparseTensor(row=row, col=col, value=val, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
t_csc, (colptr, row_csc, val_csc) = gpu_ms(lambda: SparseTensor(row=row, col=col, value=val, sparse_sizes=(M, M),
                                                                 is_sorted=True, trust_data=True).csc())
col_of = ops.ptr2ind(colptr, E)
t_exp2, (keys2, vals2) = gpu_ms(lambda: ops.spspmm_expand(col_of, row_csc, val_csc, colptr, row_csc, val_csc, offsets,
                                                          ops.ptr2ind(ops.count2ptr(ops.spspmm_count(row_csc, colptr)), total),
                                                          total, -1, torch.float32))
t_sort2, (skeys2, _) = gpu_ms(lambda: ops.sort_pairs_field(keys2, vals2, 32, M))
t_uniq2, _ = gpu_ms(lambda: ops.unique_sorted(skeys2, 1 << 32))
for name, t in (("CSC view of A", t_csc), ("count+scan+owner+expand", t_exp2), ("sort on row field", t_sort2),
                ("unique (packed)", t_uniq2)):
    print(f"  {name:24s} {t:8.3f} ms")
print(f"spspmm total: {t_all:8.3f} ms  {total / t_all / 1e6:8.2f} GProducts/s")

# CPU oracle on the first rows of A (x the full B), single thread
rows = min(M, args.cpu_rows)
cut = int(np.searchsorted(index_h[0], rows))
t0 = time.perf_counter()
ref_idx, ref_val = oracle.spspmm(index_h[:, :cut], val_h[:cut], index_h, val_h, rows, M, M)
t_cpu = time.perf_counter() - t0
n_ref = ref_idx.shape[1]
same = (np.array_equal(idxC[:, :n_ref].cpu().numpy(), ref_idx)
        and np.array_equal(valC[:n_ref].cpu().numpy(), ref_val))
print(f"CPU oracle (1 thread), first {rows} rows ({cut} entries of A): {t_cpu * 1e3:.1f} ms "
      f"-> whole product ~{t_cpu * E / max(cut, 1) * 1e3:.0f} ms; GPU result identical on those rows: {same}")
