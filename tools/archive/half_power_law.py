#!/usr/bin/env python3
"""bf16 dense operands on a power-law graph (R-MAT 21 as generated): the half-width row kernel
(one wave per row, no long-row path) against widening to fp32 and taking the fp32 route the
tensor surface picks (edge ranges + hub-row copy), conversions included."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)
N, rowptr, row, col, val = rmat_graph(21, 20_000_000, dev)
B = torch.randn(N, 128, device=dev)
Bh = B.bfloat16()
a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
for reduce in ("sum", "max"):
    with torch.no_grad():
        direct = lambda: ops._spmm(reduce, rowptr, col, val, Bh, want_arg=False)[0]
        widened = lambda: a.matmul(Bh.float(), reduce).bfloat16()
        surface = lambda: a.matmul(Bh, reduce)
        x, y, z = direct(), widened(), surface()
        print(f"spmm_{reduce} bf16 on R-MAT 21: half row kernel {event_ms(direct, 10):.3f} ms | widen + fp32 surface + narrow "
              f"{event_ms(widened, 10):.3f} ms | tensor surface today {event_ms(surface, 10):.3f} ms | "
              f"max |direct - widened| {float((x.float() - y.float()).abs().max()):.3g}", flush=True)
