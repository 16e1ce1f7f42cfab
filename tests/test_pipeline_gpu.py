"""GPU: BASELINE config 5 — coalesce + transpose + ind2ptr on a power-law
(R-MAT) COO.  Reduced scale against the oracle (bit-exact indices); full size
(100 M edges, scale 24) through size-independent properties: sortedness,
idempotence, value conservation, double transpose, pointer consistency."""
import numpy as np
import pytest
import torch

from oracle import storage_oracle as so

pytestmark = pytest.mark.gpu


def rmat(scale, nedges, seed, device="cuda", a=0.57, b=0.19, c=0.19):
    g = torch.Generator(device=device).manual_seed(seed)
    row = torch.zeros(nedges, dtype=torch.int64, device=device)
    col = torch.zeros(nedges, dtype=torch.int64, device=device)
    for bit in range(scale):
        r = torch.rand(nedges, generator=g, device=device)
        right = ((r >= a) & (r < a + b)) | (r >= a + b + c)
        down = r >= a + b
        row |= down.to(torch.int64) << bit
        col |= right.to(torch.int64) << bit
    return row, col


def test_reduced_scale_vs_oracle():
    from paddle_sparse_amd import ops, transpose

    scale, n = 14, 300_000
    N = 1 << scale
    row, col = rmat(scale, n, 4)
    val = torch.randint(-4, 5, (n,), device="cuda").float()  # exact sums
    idx, v = transpose(torch.stack([row, col]), val, N, N)
    ref_idx, ref_v = so.transpose(np.stack([row.cpu().numpy(), col.cpu().numpy()]), val.cpu().numpy(), N, N)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(v.cpu().numpy(), ref_v)
    rowptr = ops.ind2ptr(idx[0].contiguous(), N)
    assert np.array_equal(rowptr.cpu().numpy(), so.Storage(ref_idx[0], ref_idx[1], None, (N, N), True).rowptr())


def test_full_size_properties():
    from paddle_sparse_amd import coalesce, ops, transpose

    scale, n = 24, 100_000_000
    N = 1 << scale
    row, col = rmat(scale, n, 4)
    val = torch.randint(-4, 5, (n,), device="cuda").float()
    index = torch.stack([row, col])
    del row, col
    idx, v = transpose(index, val, N, N)
    nnz = idx.shape[1]
    assert 0 < nnz <= n
    key = idx[0] * N + idx[1]
    assert bool((key[1:] > key[:-1]).all())            # sorted by (row, col), duplicate-free
    assert float(v.double().sum()) == float(val.double().sum())  # small integers: exact conservation
    # idempotence: coalescing the result changes nothing
    idx2, v2 = coalesce(idx, v, N, N)
    assert torch.equal(idx2, idx) and torch.equal(v2, v)
    # transposing back == coalescing the original
    back_idx, back_v = transpose(idx, v, N, N)
    orig_idx, orig_v = coalesce(index, val, N, N)
    assert torch.equal(back_idx, orig_idx) and torch.equal(back_v, orig_v)
    # pointer consistency
    rowptr = ops.ind2ptr(idx[0].contiguous(), N)
    assert int(rowptr[0]) == 0 and int(rowptr[-1]) == nnz
    assert torch.equal(ops.ptr2ind(rowptr, nnz), idx[0])
