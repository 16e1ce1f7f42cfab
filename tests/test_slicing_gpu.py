"""GPU: narrow / select / index_select / masked_select / __getitem__
 checked against dense indexing of the same matrix.
The reference's own checks for these ops are shape-only
(test/test_tensor.py:16-68)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def idx(x):
    return torch.tensor(x, dtype=torch.int64, device="cuda")


@pytest.fixture(scope="module")
def mat():
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(12)
    M, N = 70, 55
    key = np.unique(rng.integers(0, M * N, 900))
    row, col = key // N, key % N
    val = rng.standard_normal((key.size, 2)).astype(np.float32)
    t = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N))
    return t, t.to_dense().cpu().numpy()


def dense(t):
    return t.to_dense().cpu().numpy()


def check_sorted(t):
    """results must stay sorted by (row, col), like every SparseStorage"""
    row, col, _ = t.coo()
    key = (row * max(t.sparse_size(1), 1) + col).cpu().numpy()
    assert np.all(key[1:] > key[:-1])


def test_index_select_rows_cols_and_value_dim(mat):
    t, d = mat
    sel = [5, 0, 69, 5, 33]  # unsorted, with a repeat
    out = t.index_select(0, idx(sel))
    assert np.array_equal(dense(out), d[sel]) and out.sparse_sizes() == (5, 55)
    assert out.storage.rowcount().tolist() == [(np.abs(d[r]).sum(-1) > 0).sum() for r in sel]
    out = t.index_select(1, idx([54, 1, 1, 20]))
    assert np.array_equal(dense(out), d[:, [54, 1, 1, 20]])
    check_sorted(out)
    out = t.index_select(2, idx([1]))
    assert np.array_equal(dense(out), d[:, :, [1]])
    assert t.index_select(0, idx([])).nnz() == 0


def test_masked_select(mat):
    t, d = mat
    rng = np.random.default_rng(0)
    m0 = torch.from_numpy(rng.random(70) < 0.4).cuda()
    m1 = torch.from_numpy(rng.random(55) < 0.5).cuda()
    assert np.array_equal(dense(t.masked_select(0, m0)), d[m0.cpu().numpy()])
    out = t.masked_select(1, m1)
    assert np.array_equal(dense(out), d[:, m1.cpu().numpy()])
    check_sorted(out)
    keep = torch.from_numpy(rng.random(t.nnz()) < 0.5).cuda()
    sub = t.masked_select_nnz(keep, layout="coo")
    row, col, _ = t.coo()
    ref = np.zeros_like(d)
    r, c = row[keep].cpu().numpy(), col[keep].cpu().numpy()
    ref[r, c] = d[r, c]
    assert np.array_equal(dense(sub), ref)


def test_narrow_select_and_getitem(mat):
    t, d = mat
    assert np.array_equal(dense(t.narrow(0, 10, 25)), d[10:35])
    assert np.array_equal(dense(t.narrow(1, 3, 40)), d[:, 3:43])
    assert np.array_equal(dense(t.narrow(0, -5, 5)), d[-5:])
    t.storage.fill_cache_()
    n0 = t.narrow(0, 10, 25)
    assert n0.storage._rowcount is not None and n0.storage.rowptr().tolist()[0] == 0
    assert np.array_equal(dense(t.select(0, 7)), d[7:8])
    assert np.array_equal(dense(t[3:20]), d[3:20])
    assert np.array_equal(dense(t[:, 5:30]), d[:, 5:30])
    assert np.array_equal(dense(t[idx([4, 2, 2])]), d[[4, 2, 2]])
    assert np.array_equal(dense(t[..., 1:2]), d[..., 1:2])
    mask = torch.zeros(70, dtype=torch.bool, device="cuda")
    mask[[1, 8, 40]] = True
    assert np.array_equal(dense(t[mask, 10:20]), d[[1, 8, 40]][:, 10:20])
    assert np.array_equal(dense(t[np.array([6, 7])]), d[[6, 7]])
    with pytest.raises(ValueError):
        t[::2]


def test_index_select_nnz(mat):
    t, d = mat
    pick = idx([0, 5, 17])
    out = t.index_select_nnz(pick, layout="coo")
    row, col, val = t.coo()
    assert out.storage.row().tolist() == row[pick].tolist() and out.storage.col().tolist() == col[pick].tolist()
    out = t.index_select_nnz(pick, layout="csc")
    perm = t.storage.csc2csr()[pick]
    assert out.storage.col().tolist() == col[perm].tolist()
