#!/usr/bin/env python3
"""BASELINE config 4 whole on ONE MI355X (16 M x 16 M, 160 M entries, F = 256): the eight row blocks of
`partition_rows_by_nnz` each through `shard_csr` + the block's planned local kernel, timed one by one with HIP events,
beside the unsharded product.  Produces profiles/r04_c4_all_shards.txt.  (tests/test_full_size_gpu.py holds the
bit-equality of the two.)"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import ops  # noqa: E402
from paddle_sparse_amd.distributed import partition_rows_by_nnz, shard_csr  # noqa: E402
from paddle_sparse_amd.matmul import spmm_planned  # noqa: E402

Mg = Ng = 16_000_000
NNZ, F, WORLD = 160_000_000, 256, 8
g = torch.Generator(device="cuda").manual_seed(3)
row = torch.sort(torch.randint(0, Mg, (NNZ,), generator=g, device="cuda"))[0]
rowptr = ops.ind2ptr(row, Mg)
del row
col = torch.randint(0, Ng, (NNZ,), generator=g, device="cuda")
val = torch.randn(NNZ, generator=g, device="cuda")
B = torch.randn(Ng, F, generator=g, device="cuda")
out = torch.empty(Mg, F, device="cuda")


def ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


bounds = partition_rows_by_nnz(rowptr, WORLD)
print(f"config 4 whole: {Mg} x {Ng}, nnz {NNZ}, F {F} fp32; B {B.numel() * 4 / 1e9:.2f} GB, out the same; "
      f"{torch.cuda.max_memory_allocated() / 1e9:.1f} GB allocated at peak so far")
total = 0.0
for r in range(WORLD):
    shard = shard_csr(rowptr, col, val, Ng, bounds, r)
    st = shard.storage()
    t = ms(lambda: spmm_planned(st, shard.value, B, "sum", out=out[bounds[r]:bounds[r + 1]]))
    alg = shard.nnz * (8 + 4 + 4 * F) + shard.num_rows * (8 + 4 * F)
    total += t
    print(f"  block {r}: rows [{bounds[r]}, {bounds[r + 1]}) = {shard.num_rows}, nnz {shard.nnz} "
          f"({100 * shard.nnz / NNZ:.3f} %), plan {st._spmm_algo()}: {t:.3f} ms = {shard.nnz / t / 1e6:.2f} GEdges/s, "
          f"{alg / 1e9:.2f} GB algorithmic -> {alg / t / 1e9:.2f} TB/s = {100 * alg / t / 1e9 / 8:.1f} % of 8 TB/s")
    del shard, st
t_whole = ms(lambda: ops._spmm("sum", rowptr, col, val, B, out=out), reps=5)
alg = NNZ * (8 + 4 + 4 * F) + Mg * (8 + 4 * F)
print(f"sum of the 8 blocks: {total:.3f} ms; the unsharded product in one launch: {t_whole:.3f} ms = "
      f"{NNZ / t_whole / 1e6:.2f} GEdges/s, {alg / 1e9:.1f} GB algorithmic -> {alg / t_whole / 1e9:.2f} TB/s = "
      f"{100 * alg / t_whole / 1e9 / 8:.1f} % of 8 TB/s")
