#!/usr/bin/env python3
"""Stress the single-sweep sort's inter-workgroup look-back under uneven load:
many sorts of odd sizes and key widths run back to back while a second stream
keeps the chip busy with SpMM launches; every result is compared with
torch.sort(stable=True) element by element."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_workload  # noqa: E402
from paddle_sparse_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rowptr, col, val = make_workload(500_000, 500_000, 5_000_000, 64, 3, dev)
B = torch.randn(500_000, 64, device=dev)
side = torch.cuda.Stream()
sizes = [1, 63, 8191, 8192, 8193, 100_003, 1_000_000, 7_777_777, 33_333_333]
bits = [1, 7, 8, 9, 24, 33, 48, 62]
bad = 0
runs = 0
for rep in range(3):
    for n in sizes:
        for b in bits:
            keys = torch.randint(0, 1 << b, (n,), generator=g, device=dev)
            if rep == 1:
                keys = torch.sort(keys)[0]          # sorted input: degenerate digits
            if rep == 2:
                keys[: n // 2] = keys[0]            # half of the keys identical
            with torch.cuda.stream(side):            # uneven background load
                for _ in range(3):
                    ops.spmm_sum(rowptr, col, val, B)
            srt, perm = ops.index_sort(keys, 1 << b, with_sorted_inputs=True)
            ts, tp = torch.sort(keys, stable=True)
            ok = bool(torch.equal(srt, ts)) and bool(torch.equal(perm, tp))
            runs += 1
            if not ok:
                bad += 1
                print("MISMATCH", rep, n, b)
torch.cuda.synchronize()
print(f"{runs} sorts, {bad} mismatches")
sys.exit(1 if bad else 0)
