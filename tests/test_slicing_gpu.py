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


# ---- against the oracle's restatement of index_select.py / masked_select.py / narrow.py -------------

def _same(t, ref, caches=None):
    """Same entries, order, values and sizes as the oracle's Storage; caches the reference
    passes to the new SparseStorage hold the same numbers here whenever they are present."""
    row, col, val = t.coo()
    assert t.sparse_sizes() == (ref.M, ref.N)
    assert np.array_equal(row.cpu().numpy(), ref.row) and np.array_equal(col.cpu().numpy(), ref.col)
    assert (val is None) == (ref.value is None)
    if val is not None:
        assert np.array_equal(val.cpu().numpy(), ref.value)
    for name, want in (caches or {}).items():
        got = getattr(t.storage, "_" + name)
        if got is not None:
            assert np.array_equal(got.cpu().numpy(), want), name


@pytest.fixture(scope="module")
def pair():
    from oracle import storage_oracle as so
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(31)
    M, N = 70, 55
    key = np.unique(rng.integers(0, M * N, 900))
    row, col = key // N, key % N
    val = rng.standard_normal((key.size, 3)).astype(np.float32)
    t = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N))
    return t, so.Storage(row, col, val, (M, N), is_sorted=True)


def test_slicing_vs_oracle(pair):
    from oracle import storage_oracle as so

    t, st = pair
    rng = np.random.default_rng(5)
    for sel in ([5, 0, 69, 5, 33], [], list(range(70))):
        _same(t.index_select(0, idx(sel)), *so.index_select(st, 0, sel))
    for sel in ([54, 1, 1, 20], [0], list(range(54, -1, -1))):
        _same(t.index_select(1, idx(sel)), *so.index_select(st, 1, sel))
    _same(t.index_select(2, idx([2, 0])), *so.index_select(st, 2, [2, 0]))
    _same(t.index_select(-1, idx([1])), *so.index_select(st, -1, [1]))
    pick = rng.integers(0, st.row.size, 40)
    for layout in ("coo", "csc"):
        _same(t.index_select_nnz(idx(pick), layout=layout), so.index_select_nnz(st, pick, layout))
    m0, m1, m2 = rng.random(70) < 0.4, rng.random(55) < 0.5, np.array([True, False, True])
    _same(t.masked_select(0, torch.from_numpy(m0).cuda()), *so.masked_select(st, 0, m0))
    _same(t.masked_select(1, torch.from_numpy(m1).cuda()), *so.masked_select(st, 1, m1))
    _same(t.masked_select(2, torch.from_numpy(m2).cuda()), *so.masked_select(st, 2, m2))
    keep = rng.random(st.row.size) < 0.5
    for layout in ("coo", "csc"):
        _same(t.masked_select_nnz(torch.from_numpy(keep).cuda(), layout=layout), so.masked_select_nnz(st, keep, layout))
    for dim, start, length in ((0, 10, 25), (0, -5, 5), (0, 0, 70), (1, 3, 40), (1, -10, 10), (2, 1, 2), (-1, 0, 1)):
        _same(t.narrow(dim, start, length), *so.narrow(st, dim, start, length))
    t.storage.fill_cache_()  # with every cache present the narrowed caches are the reference's slices
    for dim, start, length in ((0, 10, 25), (1, 3, 40)):
        _same(t.narrow(dim, start, length), *so.narrow(st, dim, start, length))


def test_getitem_shapes_of_the_reference(kats):
    """test/test_tensor.py:16-68: the reference's own (shape-only) checks of __getitem__."""
    from paddle_sparse_amd import SparseTensor

    k = kats["getitem_shapes"]
    m, n, kk = k["m"], k["n"], k["k"]
    g = torch.Generator(device="cuda").manual_seed(1234)
    mat = SparseTensor.from_dense(torch.randn(m, n, generator=g, device="cuda"))
    idx1 = torch.randint(0, m, (kk,), generator=g, device="cuda")
    idx2 = torch.randint(0, n, (kk,), generator=g, device="cuda")
    bool1 = torch.zeros(m, dtype=torch.bool, device="cuda")
    bool2 = torch.zeros(n, dtype=torch.bool, device="cuda")
    bool1[idx1] = True
    bool2[idx2] = True
    env = dict(mat=mat, k=kk, idx1=idx1, idx2=idx2, bool1=bool1, bool2=bool2,
               idx1np=idx1.cpu().numpy(), idx2np=idx2.cpu().numpy(), bool1np=bool1.cpu().numpy(), bool2np=bool2.cpu().numpy(),
               idx1list=idx1.tolist(), idx2list=idx2.tolist(), bool1list=bool1.tolist(), bool2list=bool2.tolist())
    sizes = dict(m=m, n=n, k=kk, k1_bool=int(bool1.sum()), k2_bool=int(bool2.sum()))
    for case in k["cases"]:
        got = eval(case["expr"], {}, env)  # noqa: S307 - expressions come from the committed fixture
        assert list(got.sizes()) == [sizes[s] for s in case["sizes"]], case["expr"]
