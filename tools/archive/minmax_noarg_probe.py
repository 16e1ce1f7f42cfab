#!/usr/bin/env python3
"""spmm_max forward on config 3 with and without the int64 arg_out, and the
autograd step that results (tools/archive/spmm_suite.py has the rest)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import make_workload  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)
M, nnz, F = 2_000_000, 20_000_000, 128
rowptr, col, val = make_workload(M, M, nnz, F, 2, dev)
B = torch.randn(M, F, device=dev)
G = torch.randn(M, F, device=dev)


def ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


for name, kw in (("out + arg_out (int64)", {}), ("out + arg_out + arg_bytes", {"want_arg_bytes": True}),
                 ("out + arg_bytes", {"want_arg_bytes": True, "want_arg": False}), ("out only", {"want_arg": False})):
    print(f"spmm_max forward, {name:28s} {ms(lambda: ops._spmm('max', rowptr, col, val, B, **kw)):7.3f} ms")

ops.spmm_set_variant(19)  # out only, but with the kernel that tracks the winners' ids
print(f"spmm_max forward, out only, ids tracked anyway   {ms(lambda: ops._spmm('max', rowptr, col, val, B, want_arg=False)):7.3f} ms")
ops.spmm_set_variant(0)
print(f"spmm_sum forward (for scale)                     {ms(lambda: ops._spmm('sum', rowptr, col, val, B)):7.3f} ms")

v = val.clone().requires_grad_(True)
Bt = B.clone().requires_grad_(True)
A = SparseTensor(rowptr=rowptr, col=col, value=v, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
print("longest row:", A.storage._longest_row())


def step():
    v.grad = Bt.grad = None
    A.matmul(Bt, "max").backward(G)


print(f"autograd fwd+bwd (bytes only, chosen by the forward): {ms(step):7.3f} ms")
real = ops._spmm
ops._spmm = lambda *a, **k: real(*a, **{**k, "want_arg": True})
print(f"autograd fwd+bwd (arg_out kept):                      {ms(step):7.3f} ms")
ops._spmm = real
with torch.no_grad():
    print(f"inference A.matmul(B, 'max') under no_grad:           {ms(lambda: A.matmul(B, 'max')):7.3f} ms")
