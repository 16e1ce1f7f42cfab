#!/usr/bin/env python3
"""Forward + backward of spmm_sum / spmm_mean on R-MAT scale 21 through the tensor surface:
fixed adjacency (gradient wrt the dense operand only) and trained edge values."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseTensor, coalesce, ops  # noqa: E402

dev = torch.device("cuda", 0)
F = 128
for relabel in (False, True, "both"):
    N, rowptr, row, col, val = rmat_graph(21, 20_000_000, dev, relabel=relabel is True)
    if relabel == "both":  # Graph500-style vertex relabelling: rows and columns by one permutation
        perm = torch.randperm(N, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
        index, val = coalesce(torch.stack([perm[row], perm[col]]), val, N, N)
        row, col = index[0].contiguous(), index[1].contiguous()
        rowptr = ops.ind2ptr(row, N)
    B = torch.randn(N, F, device=dev, requires_grad=True)
    G = torch.randn(N, F, device=dev)
    name = {False: "R-MAT 21 as generated", True: "R-MAT 21, columns relabelled",
            "both": "R-MAT 21, vertices relabelled"}[relabel]
    print(f"{name}: longest row {int((rowptr[1:] - rowptr[:-1]).max())}, longest column "
          f"{int(torch.bincount(col, minlength=N).max())}", flush=True)
    for trained in (False, True):
        v = val.clone().requires_grad_(trained)
        a = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
        a.storage.csr2csc(), a.storage.csc2csr()
        for reduce in ("sum", "max"):
            def step():
                B.grad = None
                v.grad = None
                a.matmul(B, reduce).backward(G)
            step()
            ms = event_ms(step, 10)
            fwd = event_ms(lambda: a.matmul(B, reduce), 10)  # under autograd: what the backward needs is stored
            print(f"{name}: spmm_{reduce} fwd+bwd, {'trained values' if trained else 'fixed adjacency'}: {ms:.3f} ms "
                  f"(forward {fwd:.3f})", flush=True)
