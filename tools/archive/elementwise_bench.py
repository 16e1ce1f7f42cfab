#!/usr/bin/env python3
"""add / mul / to_symmetric of C3-sized SparseTensors (2 M x 2 M, 20 M entries
each) on one MI355X: the merge of the two sorted key streams against the
"concatenate and sort" form the reference spells out (add.py:30-47).

    python tools/archive/elementwise_bench.py
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402
coalesce_mod = sys.modules["paddle_sparse_amd.coalesce"]  # the package re-exports the function under this name

M, nnz = 2_000_000, 20_000_000


def make(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    key = torch.randint(0, M * M, (nnz,), device="cuda", generator=g).unique()
    val = torch.randn(key.numel(), device="cuda", generator=g)
    return SparseTensor(row=key // M, col=key % M, value=val, sparse_sizes=(M, M), is_sorted=True, trust_data=True)


def ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out


A, B = make(1), make(2)
(ra, ca, va), (rb, cb, vb) = A.coo(), B.coo()
ka, _ = ops.make_keys(ra, ca, M)
kb, _ = ops.make_keys(rb, cb, M)
t_merge, (merged, _, pay) = ms(lambda: ops.merge_sorted(ka, kb, va, vb, want_source=False))
t_sort, (skeys, spay) = ms(lambda: ops.sort_pairs(torch.cat([ka, kb]), torch.cat([va, vb]), M * M))
assert torch.equal(merged, skeys) and torch.equal(pay, spay)
n2 = ka.numel() + kb.numel()
print(f"{ka.numel()} + {kb.numel()} sorted keys (+ fp32 payload)")
print(f"  merge_sorted              {t_merge:7.3f} ms  ({n2 * 24 / t_merge / 1e6:7.0f} GB/s of 12 B in + 12 B out per key)")
print(f"  cat + sort_pairs (6 pass) {t_sort:7.3f} ms   -> identical output")

t_add, C = ms(lambda: A + B)
t_old, ref = ms(lambda: coalesce_mod._coalesce_sorted_stream(torch.cat([ra, rb]), torch.cat([ca, cb]), torch.cat([va, vb]),
                                                             M, M, "sum"))
same = torch.equal(C.storage.row(), ref[0]) and torch.equal(C.storage.col(), ref[1]) and torch.equal(C.storage.value(), ref[2])
print(f"A + B   -> {C.nnz()} entries: {t_add:7.3f} ms  (concatenate + coalesce: {t_old:7.3f} ms, identical: {same})")
t_mul, D = ms(lambda: A * B)
print(f"A * B   -> {D.nnz()} entries: {t_mul:7.3f} ms")
t_sym_cold, S = ms(lambda: SparseTensor(row=ra, col=ca, value=va, sparse_sizes=(M, M), is_sorted=True,
                                        trust_data=True).to_symmetric())
A.csc()
t_sym_warm, _ = ms(lambda: A.to_symmetric())
print(f"A.to_symmetric() -> {S.nnz()} entries: cold {t_sym_cold:7.3f} ms (builds the CSC view), CSC cached {t_sym_warm:7.3f} ms")
