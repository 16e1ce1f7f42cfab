"""GPU: sample_adj, sample, permute and cat against the
reference's known answers (test/test_sample.py, test/test_permute.py,
test/test_cat.py) and bit-exact against oracle/sample_oracle.c on seeded
graphs — all three selection branches, because draws are counter based."""
import numpy as np
import pytest
import torch

import oracle
from util import random_csr, skewed_csr

pytestmark = pytest.mark.gpu


def idx(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.int64).cuda()


def test_sample_adj_kat(kats):
    from paddle_sparse_amd import SparseTensor, sample_adj

    k = kats["sample_adj"]
    value = torch.arange(len(k["row"]), device="cuda")
    adj = SparseTensor(row=idx(k["row"]), col=idx(k["col"]), value=value, sparse_sizes=tuple(k["sparse_sizes"]))
    out, n_id = sample_adj(adj, idx(k["subset"]), num_neighbors=-1)
    a = k["all_neighbors"]
    assert n_id.tolist() == a["n_id"]
    row, col, val = out.coo()
    assert row.tolist() == a["row"] and col.tolist() == a["col"] and val.tolist() == a["val"]
    assert out.sparse_sizes() == (4, 6)
    out, n_id = adj.sample_adj(idx(k["subset"]), 2, replace=True)
    assert out.nnz() == k["nnz_2_with_replacement"]
    out, n_id = adj.sample_adj(idx(k["subset"]), 2, replace=False)
    assert out.nnz() == k["nnz_2_without_replacement"]  # node 3 has only one edge


@pytest.mark.parametrize("k,replace", [(-1, False), (5, False), (5, True), (40, False), (1, True), (0, False)])
@pytest.mark.parametrize("graph", ["uniform", "skewed", "duplicate_subset"])
def test_sample_adj_bit_exact_vs_oracle(k, replace, graph):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(17)
    if graph == "skewed":
        row, rowptr, col, _ = skewed_csr(5000, 4000, 5, long_rows=(0, 17, 4999), long_deg=3000)
        M, N = 5000, 4000
    else:
        row, rowptr, col, _ = random_csr(20_000, 20_000, 300_000, 6)
        M = N = 20_000
    subset = rng.permutation(M)[:1500]
    if graph == "skewed":
        subset[:3] = [0, 17, 4999]  # the 3000-edge rows
    if graph == "duplicate_subset":
        subset[100:200] = subset[:100]
    ref = oracle.sample_adj(rowptr, col, subset, k, replace, seed=99, num_nodes=max(M, N))
    got = ops.sample_adj(idx(rowptr), idx(col), idx(subset), k, replace, seed=99, num_cols=N)
    for name, g, r in zip(("rowptr", "col", "n_id", "e_id"), got, ref):
        assert np.array_equal(g.cpu().numpy(), r), (name, graph, k, replace)
    # without the size hint the op sizes its scratch from col.max()
    got2 = ops.sample_adj(idx(rowptr), idx(col), idx(subset), k, replace, seed=99)
    assert all(torch.equal(a, b) for a, b in zip(got, got2))


def test_sample_adj_whole_graph_and_empty_subset():
    from paddle_sparse_amd import SparseTensor, ops

    row, rowptr, col, val = random_csr(3000, 3000, 50_000, 8)
    every = np.arange(3000)
    ref = oracle.sample_adj(rowptr, col, every, -1)
    got = ops.sample_adj(idx(rowptr), idx(col), idx(every), -1)
    for g, r in zip(got, ref):
        assert np.array_equal(g.cpu().numpy(), r)
    # identity relabelling: the sampled adjacency is the graph itself
    assert np.array_equal(got[2].cpu().numpy(), every) and np.array_equal(got[0].cpu().numpy(), rowptr)
    empty = torch.empty(0, dtype=torch.int64, device="cuda")
    rp, c, n_id, e_id = ops.sample_adj(idx(rowptr), idx(col), empty, 3)
    assert rp.tolist() == [0] and c.numel() == 0 and n_id.numel() == 0 and e_id.numel() == 0
    adj = SparseTensor(rowptr=idx(rowptr), col=idx(col), value=torch.from_numpy(val).cuda(),
                       sparse_sizes=(3000, 3000), is_sorted=True)  # keep the edge order of (rowptr, col)
    out, n_id = adj.sample_adj(idx([5, 7, 9]), 4, seed=3)
    r, c, v = out.coo()
    e = oracle.sample_adj(rowptr, col, np.array([5, 7, 9]), 4, False, 3)[3]
    assert np.array_equal(v.cpu().numpy(), val[e])
    torch.manual_seed(1)
    a = adj.sample_adj(idx([5, 7, 9]), 2)[0].coo()[1]
    torch.manual_seed(1)
    b = adj.sample_adj(idx([5, 7, 9]), 2)[0].coo()[1]
    assert torch.equal(a, b)


def test_sample(kats):
    """test/test_sample.py:8-14 plus: every pick is a neighbour of its row."""
    from paddle_sparse_amd import SparseTensor, sample

    adj = SparseTensor(row=idx([0, 0, 2, 2]), col=idx([1, 2, 0, 1]), sparse_sizes=(3, 3))
    out = sample(adj, num_neighbors=1)
    assert out.shape == (3, 1) and out.min() >= 0 and out.max() <= 2
    row, rowptr, col, _ = random_csr(2000, 1500, 30_000, 9)
    adj = SparseTensor(rowptr=idx(rowptr), col=idx(col), sparse_sizes=(2000, 1500))
    subset = np.flatnonzero(rowptr[1:] > rowptr[:-1])[:800]
    picks = adj.sample(6, idx(subset)).cpu().numpy()
    assert picks.shape == (800, 6)
    for i, n in enumerate(subset[:200]):
        assert set(picks[i]) <= set(col[rowptr[n]:rowptr[n + 1]])


def test_permute_kat(kats):
    from paddle_sparse_amd import SparseTensor

    k = kats["permute"]
    adj = SparseTensor(row=idx(k["row"]), col=idx(k["col"]),
                       value=torch.tensor(k["value"], dtype=torch.float32, device="cuda"))
    row, col, value = adj.permute(idx(k["perm"])).coo()
    assert row.tolist() == k["out_row"] and col.tolist() == k["out_col"] and value.tolist() == k["out_value"]


def test_cat_kat(kats):
    from paddle_sparse_amd import SparseTensor, cat

    k = kats["cat"]
    mat1 = SparseTensor(row=idx(k["row1"]), col=idx(k["col1"]))
    mat1.fill_cache_()
    mat2 = SparseTensor(row=idx(k["row2"]), col=idx(k["col2"]))
    mat2.fill_cache_()
    out = cat([mat1, mat2], dim=0)
    assert out.to_dense().tolist() == k["dim0"]
    assert out.storage.has_row() and out.storage.has_rowptr() and out.storage.has_rowcount()
    assert out.storage.num_cached_keys() == k["dim0_cached"]
    out = cat([mat1, mat2], dim=1)
    assert out.to_dense().tolist() == k["dim1"]
    assert out.storage.has_row() and not out.storage.has_rowptr()
    assert out.storage.num_cached_keys() == k["dim1_cached"]
    out = cat([mat1, mat2], dim=(0, 1))
    assert out.to_dense().tolist() == k["diag"]
    assert out.storage.has_row() and out.storage.has_rowptr()
    assert out.storage.num_cached_keys() == k["diag_cached"]
    value = torch.randn((mat1.nnz(), 4), device="cuda")
    mat1 = mat1.set_value_(value, layout="coo")
    out = cat([mat1, mat1], dim=-1)
    assert list(out.storage.value().shape) == [mat1.nnz(), 8]
    assert out.storage.has_row() and out.storage.has_rowptr()
    assert out.storage.num_cached_keys() == 5
    with pytest.raises(IndexError):
        cat([mat1, mat1], dim=3)


def test_narrow_diag_undoes_cat_diag_and_eye():
    """__narrow_diag__ (narrow.py:103-168) is the inverse of cat(dim=(0, 1)),
    caches included; eye() is the functional identity (eye.py:6-24)."""
    from paddle_sparse_amd import SparseTensor, __narrow_diag__, cat, eye

    rng = np.random.default_rng(21)
    mats = []
    for M, N, nnz in ((30, 20, 150), (10, 45, 200), (25, 25, 90)):
        key = np.unique(rng.integers(0, M * N, nnz))
        v = rng.standard_normal(key.size).astype(np.float32)
        t = SparseTensor(row=idx(key // N), col=idx(key % N), value=torch.from_numpy(v).cuda(), sparse_sizes=(M, N))
        mats.append(t.fill_cache_())
    big = cat(mats, dim=(0, 1))
    r0 = c0 = 0
    for t in mats:
        M, N = t.sparse_sizes()
        blk = __narrow_diag__(big, (r0, c0), (M, N))
        assert blk.sparse_sizes() == (M, N)
        for name in ("_row", "_rowptr", "_col", "_value", "_rowcount", "_colptr", "_colcount", "_csr2csc", "_csc2csr"):
            assert torch.equal(getattr(blk.storage, name), getattr(t.storage, name)), name
        assert torch.equal(big.__narrow_diag__((r0, c0), (M, N)).to_dense(), t.to_dense())
        r0, c0 = r0 + M, c0 + N
    index, value = eye(5, dtype=torch.float32, device="cuda")
    assert index.tolist() == [[0, 1, 2, 3, 4]] * 2 and value.tolist() == [1.0] * 5
    assert index.dtype == torch.int64 and index.is_cuda


def test_cat_random_vs_numpy():
    from paddle_sparse_amd import SparseTensor, cat

    rng = np.random.default_rng(12)
    mats, dense = [], []
    for M, N, nnz in ((40, 30, 200), (25, 50, 300), (60, 10, 100)):
        key = np.unique(rng.integers(0, M * N, nnz))
        v = rng.standard_normal(key.size).astype(np.float32)
        mats.append(SparseTensor(row=idx(key // N), col=idx(key % N), value=torch.from_numpy(v).cuda(),
                                 sparse_sizes=(M, N)))
        d = np.zeros((M, N), np.float32)
        d[key // N, key % N] = v
        dense.append(d)
    out = cat(mats, 0).to_dense().cpu().numpy()
    ref = np.zeros((125, 50), np.float32)
    r = 0
    for d in dense:
        ref[r:r + d.shape[0], :d.shape[1]] = d
        r += d.shape[0]
    assert np.array_equal(out, ref)
    out = cat(mats, 1).to_dense().cpu().numpy()
    ref = np.zeros((60, 90), np.float32)
    c = 0
    for d in dense:
        ref[:d.shape[0], c:c + d.shape[1]] = d
        c += d.shape[1]
    assert np.array_equal(out, ref)
    out = cat(mats, (0, 1))
    ref = np.zeros((125, 90), np.float32)
    r = c = 0
    for d in dense:
        ref[r:r + d.shape[0], c:c + d.shape[1]] = d
        r, c = r + d.shape[0], c + d.shape[1]
    assert np.array_equal(out.to_dense().cpu().numpy(), ref)
    assert out.storage.is_coalesced()
