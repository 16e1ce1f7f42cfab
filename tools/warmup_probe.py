#!/usr/bin/env python3
"""Does a kernel's time drift over the first launches on fresh operands?  Consecutive windows of
10 launches of the config-3 forward, fp32 then bf16 then fp32 again (HIP events per window)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, nnz, F = 2_000_000, 20_000_000, 128
rowptr, col, val = make_workload(M, M, nnz, F, 2, dev)
B = torch.randn(M, F, device=dev)
for name, mat in (("fp32", B), ("bf16", B.to(torch.bfloat16)), ("fp32 again", B), ("bf16, new copy", B.to(torch.bfloat16))):
    ops._spmm("sum", rowptr, col, val, mat)
    torch.cuda.synchronize()
    line = f"{name:16s}"
    for w in range(8):
        line += f" {event_ms(lambda: ops._spmm('sum', rowptr, col, val, mat), 10):.3f}"
    print(line, flush=True)
