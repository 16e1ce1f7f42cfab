#!/usr/bin/env python3
"""Generates tests/golden/third_party.npz: seeded inputs and the answers of
INDEPENDENT implementations that exist in this image, for the rows of the path
the reference holds no fixture for (SpMM beyond the README
vector, reduce over dim 0 / 1, mean / min coalesce).

  spmm fwd + bwd   torch-CPU torch.sparse.mm(sparse_csr, dense, reduce) with
                   autograd (sum / mean / amax / amin), float32
  reduce dim 0/1   numpy ufunc.at / reduceat
  coalesce         torch stable argsort + torch.segment_reduce

Neither the oracle (oracle/) nor the HIP path is involved: tests/test_oracle.py
holds the oracle to these numbers, tests/test_golden_gpu.py the HIP path.
Run here (CPU only):  python tests/golden/make_golden.py
"""
from pathlib import Path

import numpy as np
import torch

out = {}
rng = np.random.default_rng(20261004)

# ---- SpMM forward + backward ---------------------------------------------------
for tag, (M, N, K, nnz_target) in {"a": (300, 200, 8, 3000), "b": (257, 131, 32, 2500), "c": (64, 500, 128, 4000)}.items():
    key = np.unique(rng.integers(0, M * N, nnz_target))
    row, col = key // N, key % N
    row[row == 5] = 6  # leave row 5 empty ...
    order = np.lexsort((col, row))
    row, col = row[order], col[order]
    keep = np.concatenate([[True], (row[1:] != row[:-1]) | (col[1:] != col[:-1])])
    row, col = row[keep], col[keep]  # ... and stay duplicate-free
    nnz = row.size
    val = rng.standard_normal(nnz).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    rowptr = np.searchsorted(row, np.arange(M + 1)).astype(np.int64)
    out.update({f"spmm_{tag}_rowptr": rowptr, f"spmm_{tag}_col": col, f"spmm_{tag}_val": val,
                f"spmm_{tag}_B": B, f"spmm_{tag}_G": G})
    for reduce, tr in (("sum", "sum"), ("mean", "mean"), ("max", "amax"), ("min", "amin")):
        v = torch.tensor(val, requires_grad=True)
        Bt = torch.tensor(B, requires_grad=True)
        A = torch.sparse_csr_tensor(torch.tensor(rowptr), torch.tensor(col), v, size=(M, N))
        o = torch.sparse.mm(A, Bt, tr)
        o.backward(torch.tensor(G))
        out[f"spmm_{tag}_{reduce}_out"] = o.detach().numpy()
        out[f"spmm_{tag}_{reduce}_gval"] = v.grad.numpy()
        out[f"spmm_{tag}_{reduce}_gmat"] = Bt.grad.numpy()

# ---- reduce over dim 0 / dim 1 ---------------------------------------------------
M, N = 120, 90
key = np.unique(rng.integers(0, M * N, 1500))
row, col = key // N, key % N
val = rng.standard_normal(key.size).astype(np.float32)
out.update({"red_row": row, "red_col": col, "red_val": val, "red_shape": np.array([M, N])})
for dim, index, size in ((1, row, M), (0, col, N)):
    cnt = np.bincount(index, minlength=size)
    s = np.zeros(size, np.float64)
    np.add.at(s, index, val.astype(np.float64))
    mx = np.full(size, -np.inf)
    np.maximum.at(mx, index, val)
    mn = np.full(size, np.inf)
    np.minimum.at(mn, index, val)
    out[f"red_dim{dim}_sum"] = s.astype(np.float32)
    out[f"red_dim{dim}_mean"] = (s / np.maximum(cnt, 1)).astype(np.float32)
    out[f"red_dim{dim}_max"] = np.where(cnt > 0, mx, 0).astype(np.float32)
    out[f"red_dim{dim}_min"] = np.where(cnt > 0, mn, 0).astype(np.float32)

# ---- coalesce with every reduction (duplicates on purpose) ---------------------------
m, n, nnz = 40, 30, 4000
row, col = rng.integers(0, m, nnz), rng.integers(0, n, nnz)
val = rng.standard_normal((nnz, 2)).astype(np.float32)
key = torch.tensor(row * n + col)
perm = torch.argsort(key, stable=True)
skey = key[perm]
heads = torch.cat([torch.tensor([True]), skey[1:] != skey[:-1]])
offsets = torch.cat([torch.nonzero(heads).flatten(), torch.tensor([nnz])])
out.update({"co_row": row, "co_col": col, "co_val": val, "co_shape": np.array([m, n]),
            "co_index": np.stack([(skey[heads] // n).numpy(), (skey[heads] % n).numpy()])})
for op, tr in (("add", "sum"), ("mean", "mean"), ("min", "min"), ("max", "max")):
    out[f"co_{op}"] = torch.segment_reduce(torch.tensor(val)[perm], tr, offsets=offsets, axis=0).numpy()

# ---- spspmm: scipy.sparse CSR @ CSR on seeded matrices (README.md:308-353 holds one 3 x 3 vector) ------------
# float64 products and sums inside scipy, results frozen in float64; the tests hold fp32 results to
# 1e-5 * (|A| @ |B|) and the index structure exactly.  Integer-valued operands give exact answers.
import scipy.sparse as sp  # noqa: E402

for tag, (m, k, n, nnzA, nnzB, integer) in {"a": (60, 50, 70, 400, 500, False), "b": (200, 33, 150, 1500, 900, False),
                                           "c": (45, 45, 45, 300, 300, True)}.items():
    def draw(rows, cols, nnz):
        key = np.unique(rng.integers(0, rows * cols, nnz))
        r, c = key // cols, key % cols
        v = (rng.integers(-4, 5, key.size).astype(np.float32) if integer
             else rng.standard_normal(key.size).astype(np.float32))
        v[v == 0] = 1  # explicit zeros would make "stored entries" ambiguous between implementations
        return r.astype(np.int64), c.astype(np.int64), v

    ra, ca, va = draw(m, k, nnzA)
    rb, cb, vb = draw(k, n, nnzB)
    A = sp.csr_matrix((va.astype(np.float64), (ra, ca)), shape=(m, k))
    Bm = sp.csr_matrix((vb.astype(np.float64), (rb, cb)), shape=(k, n))
    # The stored structure of the product is the STRUCTURAL one (every (i, j) with some A[i, c] and B[c, j] stored):
    # upstream keeps an entry whose terms cancel, scipy's csr_matmat drops it — so the structure comes from
    # |A| @ |B| (nothing cancels) and the values from scipy's A @ B read at those positions (0 where it dropped one).
    C = (A @ Bm).tocsr()
    S = (abs(A) @ abs(Bm)).tocsr()
    S.sort_indices()
    Sc = S.tocoo()
    value = np.asarray(C[Sc.row, Sc.col]).ravel()
    dense = A.toarray() @ Bm.toarray()
    assert np.allclose(value, dense[Sc.row, Sc.col], rtol=1e-12, atol=1e-12)
    out.update({f"spspmm_{tag}_shape": np.array([m, k, n]), f"spspmm_{tag}_indexA": np.stack([ra, ca]),
                f"spspmm_{tag}_valueA": va, f"spspmm_{tag}_indexB": np.stack([rb, cb]), f"spspmm_{tag}_valueB": vb,
                f"spspmm_{tag}_index": np.stack([Sc.row.astype(np.int64), Sc.col.astype(np.int64)]),
                f"spspmm_{tag}_value": value, f"spspmm_{tag}_abs": S.data.copy(),
                f"spspmm_{tag}_cancelled": np.array([int((value == 0).sum())])})

path = Path(__file__).resolve().parent / "third_party.npz"
np.savez_compressed(path, **out)
print(path, f"{path.stat().st_size / 1024:.0f} KiB,", len(out), "arrays")
