#!/usr/bin/env python3
"""sample_adj on one MI355X against the CPU oracle (csrc/cpu/sample_cpu.cpp
restated, single thread like the reference), same graph, same seed: a
mini-batch (GraphSAGE fan-out) and a whole-graph pass.

    python tests/perf/sample_bench.py --nodes 2000000 --edges 20000000
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402  (CPU baseline leg only)
from paddle_sparse_amd import ops  # noqa: E402
from util import random_csr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=2_000_000)
ap.add_argument("--edges", type=int, default=20_000_000)
args = ap.parse_args()
M = args.nodes
row, rowptr, col, _ = random_csr(M, M, args.edges, 2)
rowptr_d, col_d = torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda()
rng = np.random.default_rng(0)


def gpu_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out


print(f"graph: {M} nodes, {col.size} edges")
for S, k, replace in ((1024, 25, False), (1024, 10, True), (65536, 10, False), (M, 10, False), (M, -1, False)):
    subset = rng.permutation(M)[:S]
    subset_d = torch.from_numpy(subset).cuda()
    t_gpu, got = gpu_ms(lambda: ops.sample_adj(rowptr_d, col_d, subset_d, k, replace, seed=7, num_cols=M),
                        reps=10 if S < M else 3)
    t0 = time.perf_counter()
    ref = oracle.sample_adj(rowptr, col, subset, k, replace, seed=7, num_nodes=M)
    t_cpu = (time.perf_counter() - t0) * 1e3
    same = all(np.array_equal(g.cpu().numpy(), r) for g, r in zip(got, ref))
    E = int(got[1].numel())
    print(f"subset {S:8d}  k={k:3d} replace={int(replace)}: {E:10d} picks, {int(got[2].numel()):9d} nodes | "
          f"GPU {t_gpu:8.3f} ms ({E / t_gpu / 1e3:8.1f} Mpicks/s) | CPU 1t {t_cpu:9.1f} ms | "
          f"x{t_cpu / t_gpu:7.1f} | identical: {same}")
