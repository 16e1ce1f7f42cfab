#!/usr/bin/env python3
"""bf16 / fp16 SpMM kernel variants on config 3 (psa_spmm_half_set_variant)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
M, nnz, F = 2_000_000, 20_000_000, 128
rowptr, col, val = make_workload(M, M, nnz, F, 2, dev)
B = torch.randn(M, F, device=dev)
lib = _lib.load()
for dt in (torch.bfloat16, torch.float16):
    Bh = B.to(dt)
    for variant in (0, 3, 0, 3, 2, 1):  # 3 = the row kernel without XCD mixing of its row blocks
        lib.psa_spmm_half_set_variant(variant)
        for red in ("sum", "max"):
            ops._spmm(red, rowptr, col, val, Bh)
            ms = event_ms(lambda: ops._spmm(red, rowptr, col, val, Bh), 20)
            ms_noarg = event_ms(lambda: ops._spmm(red, rowptr, col, val, Bh, want_arg=False), 20) if red == "max" else ms
            print(f"{dt} variant {variant} spmm_{red}: {ms:.3f} ms ({nnz / ms / 1e6:.2f} GEdges/s)"
                  + (f", out only {ms_noarg:.3f} ms" if red == "max" else ""), flush=True)
lib.psa_spmm_half_set_variant(0)
print("fp32 reference:", event_ms(lambda: ops.spmm_sum(rowptr, col, val, B), 20))
