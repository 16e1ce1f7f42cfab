#!/usr/bin/env python3
"""BASELINE config 1 (coalesce() of a 10k-edge random COO, 1000 x 1000, value
shapes [nnz] and [nnz, 2]) on one MI355X: latency of the whole call (it is
launch- and sync-bound at this size), next to the CPU oracle on the same input."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402  (CPU baseline leg only)
from oracle import storage_oracle as so  # noqa: E402
import paddle_sparse_amd as ps  # noqa: E402

rng = np.random.default_rng(0)
M = N = 1000
nnz = 10_000
row, col = rng.integers(0, M, nnz), rng.integers(0, N, nnz)
for shape in ((nnz,), (nnz, 2)):
    val = rng.standard_normal(shape).astype(np.float32)
    index_d = torch.from_numpy(np.stack([row, col])).cuda()
    val_d = torch.from_numpy(val).cuda()
    for op in ("add", "max"):
        # steady state: the call reads its count back itself, so it returns with the result
        # complete; the device-wide synchronize stays inside the timed region all the same
        for _ in range(200):
            out = ps.coalesce(index_d, val_d, M, N, op)
        torch.cuda.synchronize()
        ts = []
        for _ in range(1000):
            t0 = time.perf_counter()
            out = ps.coalesce(index_d, val_d, M, N, op)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t_gpu = float(np.median(ts)) * 1e6
        t0 = time.perf_counter()
        for _ in range(20):
            ref_i, ref_v = oracle.coalesce_c(row, col, val, M, N, op, 1)
        t_c = (time.perf_counter() - t0) / 20 * 1e6
        t0 = time.perf_counter()
        np_i, np_v = so.coalesce(np.stack([row, col]), val, M, N, op)
        t_np = (time.perf_counter() - t0) * 1e6
        ok = np.array_equal(out[0].cpu().numpy(), ref_i) and np.allclose(out[1].cpu().numpy(), ref_v, rtol=1e-5, atol=1e-6)
        print(f"coalesce 10k edges value{list(shape)} op={op}: GPU {t_gpu:7.1f} us | CPU C oracle 1t {t_c:7.1f} us | "
              f"numpy oracle {t_np:9.1f} us | {ref_i.shape[1]} entries, identical index + values within 1e-5: {ok}")
