"""GPU: empty and degenerate inputs through every op (the reference tests
empty input for ind2ptr/ptr2ind, test/test_storage.py:28-32; the same must
hold for the ops added around them)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

E64 = dict(dtype=torch.int64, device="cuda")


def empty_idx():
    return torch.empty(0, **E64)


def test_empty_through_ops():
    from paddle_sparse_amd import ops

    assert ops.ind2ptr(empty_idx(), 0).tolist() == [0]
    assert ops.ptr2ind(torch.zeros(1, **E64), 0).numel() == 0
    assert ops.gather_rows(torch.randn(5, 3, device="cuda"), empty_idx()).shape == (0, 3)
    assert ops.invert_permutation(empty_idx()).numel() == 0
    keys, flag = ops.make_keys(empty_idx(), empty_idx(), 7, check_sorted=True)
    assert keys.numel() == 0 and int(flag.item()) == 0
    count, ptr, row, col = ops.unique_sorted(empty_idx(), 5)
    assert count == 0 and ptr.tolist() == [0] and row.numel() == 0 and col.numel() == 0
    assert ops.bincount(empty_idx(), 4).tolist() == [0, 0, 0, 0]
    assert ops.bincount(empty_idx(), 0).numel() == 0
    assert ops.count2ptr(empty_idx()).tolist() == [0]
    assert ops.segment_csr(torch.empty(0, 2, device="cuda"), torch.zeros(1, **E64), "sum").shape == (0, 2)
    assert ops.segment_csr(torch.empty(0, device="cuda"), torch.zeros(4, **E64), "max").tolist() == [0, 0, 0]
    assert ops.scatter(torch.empty(0, 2, device="cuda"), empty_idx(), 3, "mean").tolist() == [[0, 0]] * 3
    out = ops.spmm_sum(torch.zeros(1, **E64), empty_idx(), None, torch.randn(4, 8, device="cuda"))
    assert out.shape == (0, 8)
    out, arg = ops.spmm_max(torch.zeros(3, **E64), empty_idx(), None, torch.empty(0, 8, device="cuda"))
    assert out.tolist() == [[0.0] * 8] * 2 and arg.tolist() == [[0] * 8] * 2
    gv = ops.spmm_value_bw(None, torch.zeros(3, **E64), empty_idx(), torch.randn(4, 8, device="cuda"),
                           torch.randn(2, 8, device="cuda"))
    assert gv.numel() == 0


def test_empty_matrix_through_the_api():
    from paddle_sparse_amd import SparseStorage, SparseTensor, coalesce, transpose

    st = SparseStorage(row=empty_idx(), col=empty_idx(), sparse_sizes=(3, 4))
    st.fill_cache_()
    assert st.rowptr().tolist() == [0, 0, 0, 0] and st.colptr().tolist() == [0] * 5
    assert st.rowcount().tolist() == [0, 0, 0] and st.colcount().tolist() == [0] * 4
    assert st.csr2csc().numel() == 0 and st.csc2csr().numel() == 0
    assert st.is_coalesced() and st.coalesce() is st
    t = SparseTensor.from_storage(st)
    assert t.t().sparse_sizes() == (4, 3) and t.nnz() == 0
    assert t.sum(0).tolist() == [0] * 4 and t.sum(1).tolist() == [0] * 3
    v = SparseTensor(row=empty_idx(), col=empty_idx(), value=torch.empty(0, device="cuda"), sparse_sizes=(3, 4))
    assert v.sum(1).tolist() == [0] * 3 and v.max(0).tolist() == [0] * 4
    out = v @ torch.randn(4, 5, device="cuda")
    assert out.shape == (3, 5) and bool((out == 0).all())
    idx, val = transpose(torch.empty((2, 0), **E64), torch.empty(0, device="cuda"), 3, 4)
    assert idx.shape == (2, 0)
    idx, val = coalesce(torch.empty((2, 0), **E64), None, 3, 4)
    assert idx.shape == (2, 0) and val is None


def test_single_entry_and_single_row_column():
    from paddle_sparse_amd import SparseTensor

    a = SparseTensor(row=torch.tensor([0], **E64), col=torch.tensor([0], **E64),
                     value=torch.tensor([2.5], device="cuda"), sparse_sizes=(1, 1))
    assert (a @ torch.tensor([[4.0]], device="cuda")).tolist() == [[10.0]]
    assert a.t().to_dense().tolist() == [[2.5]]
    # one row holding every entry (nnz > 64: several chunks per wave)
    n = 1000
    col = torch.arange(n, **E64)
    a = SparseTensor(row=torch.zeros(n, **E64), col=col, value=torch.ones(n, device="cuda"), sparse_sizes=(1, n))
    B = torch.arange(n, dtype=torch.float32, device="cuda").unsqueeze(1).repeat(1, 4)
    assert (a @ B).tolist() == [[float(n * (n - 1) // 2)] * 4]
    out = a.matmul(B, "max")
    assert out.tolist() == [[float(n - 1)] * 4]
    # one column holding every entry
    at = a.t()
    assert at.sparse_sizes() == (n, 1) and at.storage.rowcount().tolist() == [1] * n
    assert (at @ torch.ones(1, 2, device="cuda")).tolist() == [[1.0, 1.0]] * n


def test_all_duplicates_coalesce_to_one():
    from paddle_sparse_amd import coalesce

    n = 100_000
    index = torch.stack([torch.full((n,), 3, **E64), torch.full((n,), 5, **E64)])
    value = torch.ones(n, 2, device="cuda")
    for op, expect in (("add", float(n)), ("mean", 1.0), ("max", 1.0), ("min", 1.0)):
        idx, val = coalesce(index, value, 8, 8, op)
        assert idx.tolist() == [[3], [5]] and val.tolist() == [[expect, expect]]


def test_large_key_range_48_bits():
    """Matrix sides of 2^24 (BASELINE config 5's key width): keys up to 2^48."""
    from paddle_sparse_amd import coalesce

    N = 1 << 24
    rng = np.random.default_rng(0)
    row = rng.integers(0, N, 200_000)
    col = rng.integers(0, N, 200_000)
    row[:1000] = N - 1
    col[:1000] = N - 1  # duplicates at the very top of the key range
    val = np.ones(200_000, np.float32)
    idx, out = coalesce(torch.stack([torch.from_numpy(row).cuda(), torch.from_numpy(col).cuda()]),
                        torch.from_numpy(val).cuda(), N, N)
    key = np.unique(row.astype(object) * N + col.astype(object))
    assert idx.shape[1] == key.size
    got = idx[0].cpu().numpy().astype(object) * N + idx[1].cpu().numpy().astype(object)
    assert np.array_equal(got, key)
    assert float(out.sum()) == 200_000.0 and float(out[-1]) == 1000.0
