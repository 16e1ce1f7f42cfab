"""GPU parity of the host mirror (SparseStorage / SparseTensor / coalesce /
transpose / reduce / spmm) against the reference's own known-answer tests
(tests/golden/reference_kats.json) and the numpy oracle on seeded inputs.
Mirrors test/test_storage.py, test_coalesce.py, test_transpose.py,
test_reduce.py of the reference; dtypes as paddle_sparse/testing.py:12."""
import numpy as np
import pytest
import torch

import oracle
from oracle import storage_oracle as so

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.float32, torch.float64, torch.int32, torch.int64, torch.bfloat16]
DEV = "cuda"


def tensor(x, dtype):
    return None if x is None else torch.tensor(x, dtype=dtype, device=DEV)


def idx(x):
    return torch.tensor(x, dtype=torch.int64, device=DEV)


# ---- test/test_storage.py ---------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_storage(kats, dtype):
    from paddle_sparse_amd import SparseStorage

    k = kats["storage_sort"]
    st = SparseStorage(row=idx(k["row"]), col=idx(k["col"]), value=tensor(k["value"], dtype))
    assert st.row().tolist() == k["out_row"]
    assert st.col().tolist() == k["out_col"]
    assert torch.equal(st.value(), tensor(k["out_value"], dtype))
    assert st.sparse_sizes() == tuple(k["sparse_sizes"])


def test_caching(kats):
    from paddle_sparse_amd import SparseStorage

    k = kats["storage_caching"]
    row, col = idx(k["row"]), idx(k["col"])
    st = SparseStorage(row=row, col=col)
    assert st._row.tolist() == k["row"] and st._col.tolist() == k["col"] and st._value is None
    assert st._rowcount is None and st._rowptr is None and st._colcount is None
    assert st._colptr is None and st._csr2csc is None and st.num_cached_keys() == 0

    st.fill_cache_()
    for name in ("rowcount", "rowptr", "colcount", "colptr", "csr2csc", "csc2csr"):
        assert getattr(st, "_" + name).tolist() == k[name], name
    assert st.num_cached_keys() == 5

    st = SparseStorage(row=row, rowptr=st._rowptr, col=col, value=st._value,
                       sparse_sizes=st._sparse_sizes, rowcount=st._rowcount, colptr=st._colptr,
                       colcount=st._colcount, csr2csc=st._csr2csc, csc2csr=st._csc2csr)
    for name in ("rowcount", "rowptr", "colcount", "colptr", "csr2csc", "csc2csr"):
        assert getattr(st, "_" + name).tolist() == k[name], name
    assert st.num_cached_keys() == 5

    st.clear_cache_()
    assert st._rowcount is None and st._rowptr is not None and st._colcount is None
    assert st._colptr is None and st._csr2csc is None and st.num_cached_keys() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_utility(kats, dtype):
    from paddle_sparse_amd import SparseStorage

    k = kats["storage_set_value_csc"]
    value = tensor(k["value"], dtype)
    st = SparseStorage(row=idx(k["row"]), col=idx(k["col"]), value=value)
    assert st.has_value()
    st.set_value_(value, layout="csc")
    assert torch.equal(st.value(), tensor(k["csc_value"], dtype))
    st.set_value_(value, layout="coo")
    assert torch.equal(st.value(), tensor(k["coo_value"], dtype))
    st = st.set_value(value, layout="csc")
    assert torch.equal(st.value(), tensor(k["csc_value"], dtype))
    st = st.set_value(value, layout="coo")
    assert torch.equal(st.value(), tensor(k["coo_value"], dtype))

    st = st.sparse_resize((3, 3))
    assert st.sparse_sizes() == (3, 3)
    new = st.copy()
    assert new is not st and new.col().data_ptr() == st.col().data_ptr()
    new = st.clone()
    assert new is not st and new.col().data_ptr() != st.col().data_ptr()


@pytest.mark.parametrize("dtype", DTYPES)
def test_storage_coalesce(kats, dtype):
    from paddle_sparse_amd import SparseStorage

    k = kats["storage_coalesce"]
    st = SparseStorage(row=idx(k["row"]), col=idx(k["col"]), value=tensor(k["value"], dtype))
    assert st.row().tolist() == k["row"] and st.col().tolist() == k["col"]
    assert not st.is_coalesced()
    st = st.coalesce()
    assert st.is_coalesced()
    assert st.row().tolist() == k["out_row"] and st.col().tolist() == k["out_col"]
    assert torch.equal(st.value(), tensor(k["out_value"], dtype))


def test_sparse_reshape(kats):
    from paddle_sparse_amd import SparseStorage

    k = kats["storage_reshape"]
    st = SparseStorage(row=idx(k["row"]), col=idx(k["col"]))
    for step in k["steps"]:
        st = st.sparse_reshape(step["num_rows"], step["num_cols"])
        assert st.sparse_sizes() == tuple(step["sizes"])
        assert st.row().tolist() == step["row"] and st.col().tolist() == step["col"]


def test_resize_keeps_pointer_caches_consistent():
    from paddle_sparse_amd import SparseStorage

    st = SparseStorage(row=idx([0, 0, 1, 1]), col=idx([0, 1, 0, 1])).fill_cache_()
    big = st.sparse_resize((4, 3))
    assert big.rowptr().tolist() == [0, 2, 4, 4, 4] and big.colptr().tolist() == [0, 2, 4, 4]
    assert big.rowcount().tolist() == [2, 2, 0, 0] and big.colcount().tolist() == [2, 2, 0]


# ---- test/test_coalesce.py ----------------------------------------------------
def test_coalesce(kats):
    from paddle_sparse_amd import coalesce

    k = kats["coalesce"]
    index = torch.stack([idx(k["row"]), idx(k["col"])])
    out_index, _ = coalesce(index, None, m=k["m"], n=k["n"])
    assert out_index.tolist() == k["out_index"]
    for dtype in DTYPES:
        out_index, value = coalesce(index, tensor(k["value"], dtype), m=k["m"], n=k["n"])
        assert out_index.tolist() == k["out_index"]
        assert torch.equal(value, tensor(k["out_add"], dtype))
        out_index, value = coalesce(index, tensor(k["value"], dtype), m=k["m"], n=k["n"], op="max")
        assert out_index.tolist() == k["out_index"]
        assert torch.equal(value, tensor(k["out_max"], dtype))


@pytest.mark.parametrize("op", ["add", "mean", "min", "max"])
@pytest.mark.parametrize("shape", [(), (2,)])
def test_coalesce_config1_vs_oracle(op, shape):
    """BASELINE config 1: 10k-edge random COO (duplicates), fp32."""
    from paddle_sparse_amd import coalesce

    rng = np.random.default_rng(0)
    M = N = 1000
    row, col = rng.integers(0, M, 10000), rng.integers(0, N, 10000)
    val = rng.standard_normal((10000,) + shape).astype(np.float32)
    ref_idx, ref_val = so.coalesce(np.stack([row, col]), val, M, N, op)
    got_idx, got_val = coalesce(torch.stack([idx(row), idx(col)]), torch.from_numpy(val).cuda(), M, N, op)
    assert np.array_equal(got_idx.cpu().numpy(), ref_idx)  # bit-exact indices
    np.testing.assert_allclose(got_val.cpu().numpy(), ref_val, rtol=1e-5, atol=1e-6)


def test_coalesce_already_sorted_and_empty():
    from paddle_sparse_amd import coalesce

    index = torch.stack([idx([0, 0, 1]), idx([0, 2, 1])])
    value = tensor([1.0, 2.0, 3.0], torch.float32)
    out_index, out_value = coalesce(index, value, 2, 3)
    assert out_index.tolist() == [[0, 0, 1], [0, 2, 1]] and out_value.tolist() == [1, 2, 3]
    empty = torch.empty((2, 0), dtype=torch.int64, device=DEV)
    out_index, out_value = coalesce(empty, torch.empty(0, device=DEV), 4, 4)
    assert out_index.shape == (2, 0) and out_value.numel() == 0


# ---- test/test_transpose.py -----------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose(kats, dtype):
    from paddle_sparse_amd import transpose

    for name in ("transpose_matrix", "transpose"):
        k = kats[name]
        index = torch.stack([idx(k["row"]), idx(k["col"])])
        out_index, value = transpose(index, tensor(k["value"], dtype), m=k["m"], n=k["n"])
        assert out_index.tolist() == k["out_index"]
        assert torch.equal(value, tensor(k["out_value"], dtype))


def test_t_matches_oracle_and_swaps_caches():
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(3)
    M, N, nnz = 300, 200, 4000
    key = np.unique(rng.integers(0, M * N, nnz))
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    a = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N))
    a.fill_cache_()
    at = a.t()
    ref = so.t(so.Storage(row, col, val, (M, N)))
    r, c, v = at.coo()
    assert np.array_equal(r.cpu().numpy(), ref.row) and np.array_equal(c.cpu().numpy(), ref.col)
    assert np.array_equal(v.cpu().numpy(), ref.value)
    assert at.sparse_sizes() == (N, M)
    assert torch.equal(at.storage.rowptr(), a.storage.colptr())
    assert torch.equal(at.storage._colptr, a.storage.rowptr())
    assert torch.equal(at.storage._csr2csc, a.storage.csc2csr())
    assert torch.equal(at.t().storage.col(), a.storage.col())


# ---- test/test_reduce.py (+ dim 0/1, untested upstream) -----------------------------
def test_reduce_dim_none(kats):
    from paddle_sparse_amd import SparseTensor

    k = kats["reduce_dim_none"]
    value = tensor(k["value"], torch.float32)
    t = SparseTensor(row=idx(k["row"]), col=idx(k["col"]), value=value)
    assert t.sum() == k["sum"] and t.mean() == k["mean"] and t.max() == k["max"] and t.min() == k["min"]


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("npdtype,tdtype", [(np.float32, torch.float32), (np.int64, torch.int64)])
def test_reduce_dim0_dim1_vs_oracle(reduce, npdtype, tdtype):
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(5)
    M, N, nnz = 500, 400, 6000
    key = np.unique(rng.integers(0, M * N, nnz))
    row, col = key // N, key % N
    val = rng.integers(-9, 9, (key.size, 3)).astype(npdtype)
    t = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N))
    st = so.Storage(row, col, val, (M, N))
    for dim in (0, 1, -2, -3):
        got = getattr(t, reduce)(dim).cpu().numpy()
        ref = so.reduction(st, dim % 3 if dim < 0 else dim, reduce)
        if npdtype is np.float32 and reduce == "mean":
            np.testing.assert_allclose(got, ref, rtol=1e-6)
        else:
            assert np.array_equal(got, ref), (reduce, dim)
    # with the CSC caches present, dim 0 takes the atomic-free segment path: same results
    t.storage.csr2csc()
    assert t.storage.has_colptr()
    got = getattr(t, reduce)(0).cpu().numpy()
    ref = so.reduction(st, 0, reduce)
    if npdtype is np.float32 and reduce == "mean":
        np.testing.assert_allclose(got, ref, rtol=1e-6)
    else:
        assert np.array_equal(got, ref), (reduce, "csc")
    # value-less shortcuts (reduce.py:43-58)
    t0 = SparseTensor(row=idx(row), col=idx(col), sparse_sizes=(M, N))
    st0 = so.Storage(row, col, None, (M, N))
    for dim in (None, 0, 1):
        assert np.array_equal(np.asarray(getattr(t0, reduce)(dim).cpu()), np.asarray(so.reduction(st0, dim, reduce)))


# ---- tensor-level pieces on the path ---------------------------------------------------
def test_to_symmetric_kat():
    """test/test_tensor.py:75-88."""
    from paddle_sparse_amd import SparseTensor

    row, col = idx([0, 0, 0, 1, 1]), idx([0, 1, 2, 0, 2])
    value = torch.arange(1, 6, dtype=torch.float32, device=DEV)
    mat = SparseTensor(row=row, col=col, value=value)
    assert not mat.is_symmetric()
    mat = mat.to_symmetric()
    assert mat.is_symmetric()
    assert mat.to_dense().tolist() == [[2, 6, 3], [6, 0, 5], [3, 5, 0]]


def test_csr_csc_coo_and_scipy_roundtrip():
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(9)
    M, N = 60, 45
    key = np.unique(rng.integers(0, M * N, 700))
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    t = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N))
    for layout in ("coo", "csr", "csc"):
        sp = t.to_scipy(layout)
        assert np.array_equal(sp.toarray(), t.to_dense().cpu().numpy())
    back = SparseTensor.from_scipy(t.to_scipy("csc"), device=DEV)
    assert back == t
    colptr, r_csc, v_csc = t.csc()
    ref = so.Storage(row, col, val, (M, N))
    perm = ref.csr2csc()
    assert np.array_equal(colptr.cpu().numpy(), ref.colptr())
    assert np.array_equal(r_csc.cpu().numpy(), row[perm]) and np.array_equal(v_csc.cpu().numpy(), val[perm])
    dense = SparseTensor.from_dense(t.to_dense())
    assert dense == t
    for sparse in (t.to_torch_sparse_coo_tensor(), t.to_torch_sparse_csr_tensor(), t.to_torch_sparse_csc_tensor()):
        assert torch.equal(sparse.to_dense(), t.to_dense())
    assert SparseTensor.from_torch_sparse_coo_tensor(t.to_torch_sparse_coo_tensor()) == t
    assert SparseTensor.from_torch_sparse_csr_tensor(t.to_torch_sparse_csr_tensor()) == t


def test_eye_caches():
    """test/test_eye.py:42-66."""
    from paddle_sparse_amd import SparseTensor

    for M, N in ((3, 3), (3, 4), (4, 3)):
        a = SparseTensor.eye(M, N, device=DEV, fill_cache=True)
        b = SparseTensor.eye(M, N, device=DEV, fill_cache=False).fill_cache_()
        for name in ("rowcount", "colptr", "colcount", "csr2csc", "csc2csr"):
            assert getattr(a.storage, "_" + name).tolist() == getattr(b.storage, "_" + name).tolist(), (M, N, name)
        assert a.storage.rowptr().tolist() == b.storage.rowptr().tolist()


# ---- SpMM on the tensor surface ----------------------------------------------------------
def test_spmm_functional_readme_kat(kats):
    from paddle_sparse_amd import spmm

    k = kats["spmm"]
    out = spmm(idx(k["index"]), tensor(k["value"], torch.float32), k["m"], k["n"],
               tensor(k["matrix"], torch.float32))
    assert out.tolist() == k["out"]


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("has_value", [True, False])
def test_spmm_autograd_vs_oracle(reduce, has_value):
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(21)
    M, N, K = 400, 300, 32
    key = np.unique(rng.integers(0, M * N, 5000))
    row, col = key // N, key % N
    nnz = key.size
    val = rng.standard_normal(nnz).astype(np.float32) if has_value else None
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)

    v = torch.from_numpy(val).cuda().requires_grad_() if has_value else None
    Bt = torch.from_numpy(B).cuda().requires_grad_()
    a = SparseTensor(row=idx(row), col=idx(col), value=v, sparse_sizes=(M, N))
    out = a.matmul(Bt, reduce) if reduce != "sum" else a @ Bt
    out.backward(torch.from_numpy(G).cuda())

    ref, arg = oracle.spmm(reduce, rowptr, col, val, B)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    assert np.all(np.abs(out.detach().cpu().numpy() - ref) <= 1e-5 * S + 1e-30)
    if reduce in ("sum", "mean"):
        gB = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N)
        gV = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
    else:
        gV, gB = oracle.spmm_minmax_bw(col, val, B, G, arg)
    # gradients to the north-star bar too: 1e-5 of the sum of the absolute terms
    absv = None if val is None else np.abs(val)
    if reduce in ("sum", "mean"):
        sB = oracle.spmm_mat_bw(reduce, row, rowptr, col, absv, np.abs(G), N)
        sV = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(B), np.abs(G))
    else:
        sV, sB = oracle.spmm_minmax_bw(col, absv, np.abs(B), np.abs(G), arg)
    assert np.all(np.abs(Bt.grad.cpu().numpy() - gB) <= 1e-5 * sB + 1e-30)
    if has_value:
        assert np.all(np.abs(v.grad.cpu().numpy() - gV) <= 1e-5 * sV + 1e-30)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_grad_mat_weights_are_reused_until_the_values_change(reduce):
    """The CSC-ordered weights of grad_mat are memoised on the storage for a
    fixed adjacency; an in-place change of the values must be seen."""
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(22)
    M, N, K = 300, 200, 16
    key = np.unique(rng.integers(0, M * N, 3000))
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)
    v = torch.from_numpy(val).cuda()
    a = SparseTensor(row=idx(row), col=idx(col), value=v, sparse_sizes=(M, N))

    def grad_mat():
        Bt = torch.zeros(N, K, device="cuda", requires_grad=True)
        a.matmul(Bt, reduce).backward(torch.from_numpy(G).cuda())
        return Bt.grad

    first = grad_mat()
    memo = a.storage._csc_weight_memo[3]
    again = grad_mat()
    assert a.storage._csc_weight_memo[3] is memo and torch.equal(first, again)
    scale = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(G), N)
    assert np.all(np.abs(first.cpu().numpy() - oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N))
                  <= 1e-5 * scale + 1e-30)
    v.mul_(-2.0)  # in place: same address, new version
    changed = grad_mat()
    assert a.storage._csc_weight_memo[3] is not memo
    assert np.all(np.abs(changed.cpu().numpy() - oracle.spmm_mat_bw(reduce, row, rowptr, col, -2 * val, G, N))
                  <= 1e-5 * 2 * scale + 1e-30)


def test_cpu_tensors_are_rejected_by_the_hot_path():
    from paddle_sparse_amd import SparseTensor

    t = SparseTensor(row=torch.tensor([0, 1]), col=torch.tensor([1, 0]), is_sorted=True)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        t.storage.rowptr()


def test_equal_and_to():
    """test/test_tensor.py:91-124."""
    from paddle_sparse_amd import SparseTensor

    row, value = idx([0, 0, 0, 1, 1]), torch.arange(1, 6, device=DEV)
    a = SparseTensor(row=row, col=idx([0, 1, 2, 0, 2]), value=value)
    b = SparseTensor(row=row, col=idx([0, 1, 2, 0, 2]), value=value)
    c = SparseTensor(row=row, col=idx([0, 1, 2, 0, 1]), value=value)
    assert a is not b and a == b
    assert a is not c and a != c

    assert value.dtype == torch.int64
    mat = a.to(torch.float32)
    assert mat.storage.value().dtype == torch.float32 and mat.storage.value().is_cuda
    mat = mat.to("cpu", torch.float32)
    assert not mat.storage.value().is_cuda and not mat.storage.row().is_cuda and not mat.storage.col().is_cuda
    back = mat.to(value)  # device and dtype of a tensor
    assert back.storage.value().dtype == torch.int64 and back.storage.col().is_cuda and back == a


# ---- edge values stay differentiable through every value-moving op (the reference's do) -------------
def test_value_gradients_flow_through_t_coalesce_and_friends():
    """paddle `value[perm]` / paddle_scatter.segment_csr keep autograd alive in t(), csc(),
    set_value(layout="csc"), storage.coalesce(), coalesce(), transpose(), to_symmetric() and
    index_select(); here the gathers and the segmented sum / mean are autograd Functions."""
    import paddle_sparse_amd as ps
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(33)
    M, N, K = 60, 45, 8
    key = np.unique(rng.integers(0, M * N, 700))
    row, col = key // N, key % N
    nnz = key.size
    val = rng.standard_normal(nnz).astype(np.float32)
    B = rng.standard_normal((M, K)).astype(np.float32)
    G = rng.standard_normal((N, K)).astype(np.float32)

    # (a) A.t() @ B: d/dvalue[e] = <B[row[e]], G[col[e]]>
    v = torch.from_numpy(val).cuda().requires_grad_()
    a = SparseTensor(row=idx(row), col=idx(col), value=v, sparse_sizes=(M, N))
    (a.t() @ torch.from_numpy(B).cuda()).backward(torch.from_numpy(G).cuda())
    want = (B[row] * G[col]).sum(1)
    scale = (np.abs(B[row]) * np.abs(G[col])).sum(1)
    assert np.all(np.abs(v.grad.cpu().numpy() - want) <= 1e-5 * scale + 1e-30)

    # (b) coalesce / transpose with duplicates: the gradient of the summed entry reaches every duplicate
    dup_r = np.concatenate([row, row[:100]])
    dup_c = np.concatenate([col, col[:100]])
    w = rng.standard_normal(nnz + 100).astype(np.float32)
    for fn, op in ((ps.coalesce, "add"), (ps.coalesce, "mean"), (ps.transpose, None)):
        v = torch.from_numpy(w).cuda().requires_grad_()
        index = idx(np.stack([dup_r, dup_c]))
        out_i, out_v = fn(index, v, M, N, op) if op else fn(index, v, M, N)
        coef = torch.from_numpy(rng.standard_normal(out_v.shape[0]).astype(np.float32)).cuda()
        (out_v * coef).sum().backward()
        oi = out_i.cpu().numpy()
        if op:
            pos = np.searchsorted(oi[0] * N + oi[1], dup_r * N + dup_c)
        else:  # transpose: the output is the N x M matrix, sorted by (col, row) of the input
            pos = np.searchsorted(oi[0] * M + oi[1], dup_c * M + dup_r)
        okey = oi[0]
        cnt = np.bincount(pos, minlength=okey.size)[pos]
        want = coef.cpu().numpy()[pos] / (cnt if op == "mean" else 1)
        np.testing.assert_allclose(v.grad.cpu().numpy(), want, rtol=1e-6, atol=1e-7)

    # (c) csc(), set_value(layout="csc"), to_symmetric(), index_select(), storage.coalesce()
    v = torch.from_numpy(val).cuda().requires_grad_()
    a = SparseTensor(row=idx(row), col=idx(col), value=v, sparse_sizes=(M, N))
    coef = torch.from_numpy(rng.standard_normal(nnz).astype(np.float32)).cuda()
    (a.csc()[2] * coef).sum().backward()
    perm = a.storage.csr2csc()
    assert torch.allclose(v.grad[perm], coef)
    v.grad = None
    b = a.set_value(v * 2.0, layout="csc")
    (b.storage.value() * coef).sum().backward()
    assert torch.allclose(v.grad, 2.0 * coef[a.storage.csr2csc()])
    v.grad = None
    sq = SparseTensor(row=idx(row % 40), col=idx(col % 40), value=v, sparse_sizes=(40, 40))
    sym = sq.to_symmetric()
    sym.storage.value().sum().backward()
    assert torch.allclose(v.grad, torch.full_like(v, 2.0))  # every entry appears as (i, j) and (j, i)
    v.grad = None
    sel = idx([3, 3, 10, 59])
    a.index_select(0, sel).storage.value().sum().backward()
    times = np.bincount(np.array([3, 3, 10, 59]), minlength=M)[row]
    assert np.array_equal(v.grad.cpu().numpy(), times.astype(np.float32))
    # detached paths still take the fast routes
    with torch.no_grad():
        assert ps.coalesce(idx(np.stack([dup_r, dup_c])), torch.from_numpy(w).cuda(), M, N)[1].requires_grad is False


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_grad_mat_over_a_power_law_csc_view(reduce):
    """Fixed adjacency with hub rows AND hub columns: the backward wrt the dense operand is a
    forward SpMM over the CSC view, which picks the edge-range kernels and the hub-row copy for
    itself (SparseStorage._csc_view); gradients against the oracle."""
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(44)
    M, N, K = 5000, 4000, 64
    deg = rng.integers(0, 3, M)
    deg[rng.integers(0, M, 25)] = 600
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    nnz = int(rowptr[-1])
    row = np.repeat(np.arange(M), deg)
    col = rng.integers(0, N, nnz)
    hubs = rng.integers(0, N, 30)
    pick = rng.random(nnz) < 0.6
    col[pick] = hubs[rng.integers(0, 30, int(pick.sum()))]
    key = np.unique(row * N + col)
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)
    G = rng.standard_normal((M, K)).astype(np.float32)
    a = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N), is_sorted=True)
    Bt = torch.zeros(N, K, device="cuda", requires_grad=True)
    a.matmul(Bt, reduce).backward(torch.from_numpy(G).cuda())
    view = a.storage._csc_view()
    assert view._spmm_algo() == "edge_ranges" and view.sparse_sizes() == (N, M)
    want = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N)
    scale = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(G), N)
    assert np.all(np.abs(Bt.grad.cpu().numpy() - want) <= 1e-5 * scale + 1e-30)


@pytest.mark.parametrize("reduce", ["min", "max"])
def test_minmax_fixed_adjacency_on_a_power_law_matrix(reduce):
    """Fixed edge weights, gradient wrt the dense operand only: the surface takes the masked edge-range
    pass over the CSC view (ops.spmm_minmax_bw_eb) with the hub-row copies; gradient against the oracle."""
    import paddle_sparse_amd.storage as st_mod
    from paddle_sparse_amd import SparseTensor, ops

    rng = np.random.default_rng(47)
    M, N, K = 5000, 4000, 64
    deg = rng.integers(0, 3, M)
    deg[rng.integers(0, M, 25)] = 600
    row = np.repeat(np.arange(M), deg)
    col = rng.integers(0, N, row.size)
    hubs = rng.integers(0, N, 30)
    pick = rng.random(row.size) < 0.6
    col[pick] = hubs[rng.integers(0, 30, int(pick.sum()))]
    key = np.unique(row * N + col)
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    a = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N), is_sorted=True)
    Bt = torch.from_numpy(B).cuda().requires_grad_(True)
    taken = []
    real = ops.spmm_minmax_bw_eb

    def spy(*args, **kw):
        taken.append(kw.get("hot_ids") is not None)
        return real(*args, **kw)

    old = st_mod.HOT_COLUMNS
    st_mod.HOT_COLUMNS, ops.spmm_minmax_bw_eb = 64, spy
    try:
        out = a.matmul(Bt, reduce)
        out.backward(torch.from_numpy(G).cuda())
    finally:
        st_mod.HOT_COLUMNS, ops.spmm_minmax_bw_eb = old, real
    assert taken == [True]
    ref_out, arg = oracle.spmm(reduce, rowptr, col, val, B)
    assert np.array_equal(out.detach().cpu().numpy(), ref_out)
    _, want = oracle.spmm_minmax_bw(col, val, B, G, arg, want_value=False)
    _, scale = oracle.spmm_minmax_bw(col, np.abs(val), np.abs(B), np.abs(G), arg, want_value=False)
    assert np.all(np.abs(Bt.grad.cpu().numpy() - want) <= 1e-5 * scale + 1e-30)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_trained_values_on_a_power_law_matrix(reduce):
    """Hub rows and hub columns, gradients wrt the values AND the dense operand: the one-pass
    backward over the CSC view reads the hub rows of grad_out (and of the row-local arg_out,
    two bytes per element: the longest row has more than 128 entries) from compact copies;
    both gradients against the oracle."""
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(45)
    M, N, K = 5000, 4000, 64
    deg = rng.integers(0, 3, M)
    deg[rng.integers(0, M, 25)] = 600
    nnz = int(deg.sum())
    row = np.repeat(np.arange(M), deg)
    col = rng.integers(0, N, nnz)
    hubs = rng.integers(0, N, 30)
    pick = rng.random(nnz) < 0.6
    col[pick] = hubs[rng.integers(0, 30, int(pick.sum()))]
    key = np.unique(row * N + col)
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    v = torch.from_numpy(val).cuda().requires_grad_(True)
    a = SparseTensor(row=idx(row), col=idx(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    Bt = torch.from_numpy(B).cuda().requires_grad_(True)
    import paddle_sparse_amd.storage as st_mod

    old = st_mod.HOT_COLUMNS
    st_mod.HOT_COLUMNS = 64  # (the production 65 536 never pays on a 5000-row test matrix)
    try:
        out = a.matmul(Bt, reduce)
        out.backward(torch.from_numpy(G).cuda())
    finally:
        st_mod.HOT_COLUMNS = old
    view = a.storage._csc_view()
    assert view._hot_columns() is not None and 128 < a.storage._longest_row() <= 65_536
    if reduce in ("sum", "mean"):
        want_m = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N)
        want_v = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
        scale_m = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(G), N)
        scale_v = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(B), np.abs(G))
    else:
        ref_out, arg = oracle.spmm(reduce, rowptr, col, val, B)
        assert np.array_equal(out.detach().cpu().numpy(), ref_out)
        want_v, want_m = oracle.spmm_minmax_bw(col, val, B, G, arg)
        scale_v, scale_m = oracle.spmm_minmax_bw(col, np.abs(val), np.abs(B), np.abs(G), arg)
    assert np.all(np.abs(Bt.grad.cpu().numpy() - want_m) <= 1e-5 * scale_m + 1e-30)
    assert np.all(np.abs(v.grad.cpu().numpy() - want_v) <= 1e-5 * scale_v + 1e-30)
