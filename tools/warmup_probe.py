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
import time


def windows(name, fn, n=8):
    fn()
    torch.cuda.synchronize()
    line = f"{name:44s}"
    for _ in range(n):
        line += f" {event_ms(fn, 10):.3f}"
    print(line, flush=True)


Bh = B.to(torch.bfloat16)
windows("fp32", lambda: ops._spmm("sum", rowptr, col, val, B))
windows("bf16, first use in the process", lambda: ops._spmm("sum", rowptr, col, val, Bh))
windows("fp32 again", lambda: ops._spmm("sum", rowptr, col, val, B))
windows("bf16 again (same operand, same cached out block)", lambda: ops._spmm("sum", rowptr, col, val, Bh))
# (1) is it the freshly hipMalloc'ed output block?  release the allocator's cache: the next call's `out` is new memory
torch.cuda.empty_cache()
windows("bf16 after empty_cache (out = fresh hipMalloc)", lambda: ops._spmm("sum", rowptr, col, val, Bh))
# (2) is it the clock / power state?  idle for 2 s, same blocks
torch.cuda.synchronize()
time.sleep(2.0)
windows("bf16 after 2 s idle", lambda: ops._spmm("sum", rowptr, col, val, Bh))
time.sleep(2.0)
windows("fp32 after 2 s idle", lambda: ops._spmm("sum", rowptr, col, val, B))
# (3) a caller-owned output buffer allocated long before (psa_spmm_half writes where torch.empty points)
torch.cuda.empty_cache()
keep = [torch.empty((M, F), dtype=torch.bfloat16, device=dev) for _ in range(3)]  # take fresh blocks out of the way
windows("bf16 after empty_cache + 3 held blocks", lambda: ops._spmm("sum", rowptr, col, val, Bh))
