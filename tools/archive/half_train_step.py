#!/usr/bin/env python3
"""Forward + backward with bf16 dense operands and a fixed adjacency (gradient wrt the dense
operand only), next to fp32: uniform config 3 and R-MAT 21 as generated."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, make_workload, rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)
F = 128
M = 2_000_000
rowptr, col, val = make_workload(M, M, 20_000_000, F, 0, dev)
graphs = {"uniform C3": (M, rowptr, ops.ptr2ind(rowptr, col.numel()), col, val)}
graphs["R-MAT 21 as generated"] = rmat_graph(21, 20_000_000, dev)
for name, (N, rowptr, row, col, val) in graphs.items():
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
    a.storage.csr2csc()
    for dtype in (torch.float32, torch.bfloat16):
        B = torch.randn(N, F, device=dev).to(dtype).requires_grad_()
        G = torch.randn(N, F, device=dev).to(dtype)
        for reduce in ("sum", "mean"):
            def step():
                B.grad = None
                a.matmul(B, reduce).backward(G)
            step()
            print(f"{name}: spmm_{reduce} fwd+bwd, fixed adjacency, {str(dtype).split('.')[-1]}: {event_ms(step, 10):.3f} ms", flush=True)
