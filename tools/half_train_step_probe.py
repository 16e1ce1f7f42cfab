#!/usr/bin/env python3
"""bf16 training step with TRAINED edge values on config 3: forward + backward through the tensor
surface (half-width forward, one half-width pass over the CSC view for both gradients) next to fp32."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)
M = N = 2_000_000
F, nnz = 128, 20_000_000
rowptr, col, val = make_workload(M, N, nnz, F, 2, dev)
B = torch.randn(N, F, device=dev)
G = torch.randn(M, F, device=dev)
for dtype in (torch.float32, torch.bfloat16, torch.float16):
    v = val.clone().requires_grad_()
    Bt = B.detach().to(dtype, copy=True).requires_grad_()  # a leaf of its own (B itself must stay without grad)
    Gd = G.to(dtype)
    a = SparseTensor(rowptr=rowptr, col=col, value=v, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    for reduce in ("sum", "mean", "max"):
        def step():
            v.grad = Bt.grad = None
            a.matmul(Bt, reduce).backward(Gd)
        for _ in range(25):
            step()
        print(f"{str(dtype):16s} spmm_{reduce} fwd + bwd, trained values: {event_ms(step, 20):7.3f} ms", flush=True)
    if dtype != torch.float32:
        st = a.storage
        w = ops.permute_apply(val, st._permute_plan("to_csc", force=True))
        fn = lambda: ops.spmm_half_sum_bw_csc(st.colptr(), st._row_in_csc_order(), w, Bt.detach(), Gd, True, long_columns=False)
        for _ in range(40):
            fn()
        ms = event_ms(fn, 20)
        nb = nnz * (8 + 4 + 2 * F + 4) + N * (8 + 4 * F)
        print(f"{str(dtype):16s} the half-width pass over the CSC view alone: {ms:7.3f} ms = {nb / ms / 1e6 / 8000:.3f} of 8 TB/s on {nb / 1e9:.2f} GB", flush=True)
        fn2 = lambda: ops.spmm_half_sum_bw_csc(st.colptr(), st._row_in_csc_order(), w, Bt.detach(), Gd, False, long_columns=False)
        for _ in range(40):
            fn2()
        print(f"{str(dtype):16s}   the same pass without grad_value (grad_mat only): {event_ms(fn2, 20):7.3f} ms", flush=True)
        fwd = lambda: ops._spmm("sum", st.colptr(), st._row_in_csc_order(), w, Gd)
        for _ in range(40):
            fwd()
        print(f"{str(dtype):16s}   the half-width FORWARD kernel over the same CSC view: {event_ms(fwd, 20):7.3f} ms", flush=True)
