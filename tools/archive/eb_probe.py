#!/usr/bin/env python3
"""Edge-balanced SpMM (csrc/spmm_eb.hip, variants 30-33) against the
one-wave-per-row kernels (variant 0): equality on skewed graphs at several K,
then timings on the uniform config-3 graph and on R-MAT scale 21."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import algorithmic_bytes, event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import coalesce, ops  # noqa: E402

dev = torch.device("cuda", 0)


def skewed(M, N, nnz, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    # a third of the edges in 5 hub rows, the rest uniform over half of the rows (the others stay empty)
    hubs = torch.randint(0, M, (5,), generator=g, device=dev)
    r1 = hubs[torch.randint(0, 5, (nnz // 3,), generator=g, device=dev)]
    r2 = torch.randint(0, M // 2, (nnz - nnz // 3,), generator=g, device=dev) * 2
    row = torch.sort(torch.cat([r1, r2])).values
    col = torch.randint(0, N, (nnz,), generator=g, device=dev)
    val = torch.randn(nnz, generator=g, device=dev)
    return row, ops.ind2ptr(row, M), col, val


def check():
    bad = 0
    for (M, N, nnz, K) in ((3000, 2000, 40000, 128), (3000, 2000, 40000, 64), (3000, 2000, 40000, 16),
                           (3000, 2000, 40000, 32), (1000, 700, 9000, 160), (1000, 700, 9000, 256),
                           (1000, 700, 9000, 512), (1000, 700, 9000, 8), (50, 40, 3, 128), (50, 40, 0, 128),
                           (200000, 100000, 1500000, 128)):
        row, rowptr, col, val = skewed(M, N, nnz, M + K)
        B = torch.randn(N, K, device=dev)
        B[torch.randint(0, N, (N // 4,), device=dev)] = 1.0  # ties for min/max
        for give_row in (True, False):
            for v in (None, val):
                for red in ("sum", "mean", "min", "max"):
                    ops.spmm_set_variant(0)
                    ref = ops._spmm(red, rowptr, col, v, B, want_arg_bytes=True)
                    for variant in (30, 31):
                        ops.spmm_set_variant(variant)
                        got = ops._spmm(red, rowptr, col, v, B, want_arg_bytes=True, row=row if give_row else None)
                        got2 = ops._spmm(red, rowptr, col, v, B, want_arg_bytes=True, row=row if give_row else None)
                        ok = torch.equal(got[0], got2[0])
                        if red in ("sum", "mean"):
                            w = torch.ones(nnz, device=dev) if v is None else v.abs()
                            S = ops._spmm("sum", rowptr, col, w, B.abs())[0]
                            ok &= bool(((got[0] - ref[0]).abs() <= 1e-5 * S + 1e-30).all())
                        else:
                            ok &= torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
                            deg = rowptr[1:] - rowptr[:-1]
                            ne = deg > 0
                            ok &= torch.equal(got[2][ne], ref[2][ne])
                            noarg = ops._spmm(red, rowptr, col, v, B, want_arg=False, row=row if give_row else None)
                            ok &= torch.equal(noarg[0], ref[0])
                        if not ok:
                            bad += 1
                            print(f"MISMATCH M={M} nnz={nnz} K={K} red={red} variant={variant} row={give_row} val={v is not None}")
    ops.spmm_set_variant(0)
    print("equality check:", "OK" if bad == 0 else f"{bad} mismatches")
    return bad == 0


def rmat(scale, n, seed=4):
    N = 1 << scale
    g = torch.Generator(device=dev).manual_seed(seed)
    row = torch.zeros(n, dtype=torch.int64, device=dev)
    col = torch.zeros(n, dtype=torch.int64, device=dev)
    for bit in range(scale):
        r = torch.rand(n, generator=g, device=dev)
        right = ((r >= 0.57) & (r < 0.76)) | (r >= 0.95)
        down = r >= 0.76
        row |= down.to(torch.int64) << bit
        col |= right.to(torch.int64) << bit
    index, val = coalesce(torch.stack([row, col]), torch.randn(n, generator=g, device=dev), N, N)
    return N, index[0].contiguous(), index[1].contiguous(), val


def timings():
    F = 128
    graphs = []
    M, nnz = 2_000_000, 20_000_000
    rowptr, col, val = make_workload(M, M, nnz, F, 2, dev)
    graphs.append(("uniform C3", M, rowptr, ops.ptr2ind(rowptr, nnz), col, val))
    N, row, col2, val2 = rmat(21, 20_000_000)
    graphs.append(("R-MAT 21", N, ops.ind2ptr(row, N), row, col2, val2))
    for name, M, rowptr, row, col, val in graphs:
        nnz = col.numel()
        B = torch.randn(M, F, device=dev)
        for variant, label in ((0, "row waves (round 1)"), (30, "edge ranges 256"), (31, "edge ranges 128"),
                               (32, "edge ranges 512"), (33, "edge ranges 1024")):
            ops.spmm_set_variant(variant)
            for op in ("spmm_sum", "spmm_max"):
                fn = getattr(ops, op)
                for r in ((row, None) if variant >= 30 else (None,)):
                    fn(rowptr, col, val, B, row=r)
                    ms = event_ms(lambda: fn(rowptr, col, val, B, row=r), 20)
                    alg = algorithmic_bytes(nnz, M, F, True, op == "spmm_max")
                    print(f"{name:11s} [{label:20s}] {op} row={'given' if r is not None else 'none '}: {ms:.3f} ms  "
                          f"{nnz / ms / 1e6:.2f} GEdges/s  {alg / ms / 1e9:.2f} TB/s algorithmic", flush=True)
        if name.startswith("uniform"):
            ops.spmm_set_variant(30)
            print("noarg max:", event_ms(lambda: ops._spmm("max", rowptr, col, val, B, want_arg=False, row=row), 20))
    ops.spmm_set_variant(0)


if __name__ == "__main__":
    ok = check()
    if ok and "--no-time" not in sys.argv:
        timings()
    sys.exit(0 if ok else 1)
