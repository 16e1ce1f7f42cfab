#!/usr/bin/env python3
"""coalesce() at one size, N times (for rocprofv3 --kernel-trace --stats): tools/archive/coalesce_prof.py <nnz> [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import paddle_sparse_amd as ps  # noqa: E402

nnz = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
g = torch.Generator(device="cuda").manual_seed(0)
m = n = max(int((10 * nnz) ** 0.5), 2)
index = torch.stack([torch.randint(0, m, (nnz,), generator=g, device="cuda"),
                     torch.randint(0, n, (nnz,), generator=g, device="cuda")])
value = torch.randn(nnz, generator=g, device="cuda")
for _ in range(reps):
    out = ps.coalesce(index, value, m, n)
torch.cuda.synchronize()
print(nnz, out[0].shape)
