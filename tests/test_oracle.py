"""CPU suite: pin the oracle (oracle/) against the reference's known answers
and against independent third-party implementations available in this image
(numpy, scipy, torch-CPU torch.sparse.mm).  No GPU, no HIP compute."""
import numpy as np
import pytest
import scipy.sparse
import torch

import oracle
from oracle import storage_oracle as so
from util import random_csr, skewed_csr

from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


# ---- ind2ptr / ptr2ind: reference KATs test/test_storage.py:20-32 ---------
def test_ind2ptr_kats(kats):
    for c in kats["ind2ptr"]["cases"]:
        assert oracle.ind2ptr(c["row"], c["M"]).tolist() == c["rowptr"]


def test_ptr2ind_kats(kats):
    for c in kats["ptr2ind"]["cases"]:
        assert oracle.ptr2ind(c["rowptr"], c["E"]).tolist() == c["row"]


@pytest.mark.parametrize("M,nnz,seed", [(1, 1, 0), (7, 0, 1), (100, 1000, 2), (5000, 300, 3), (1, 500, 4)])
def test_ind2ptr_vs_searchsorted(M, nnz, seed):
    row, rowptr, _, _ = random_csr(M, 10, nnz, seed)
    got = oracle.ind2ptr(row, M)
    assert np.array_equal(got, rowptr)
    assert np.array_equal(oracle.ptr2ind(got, nnz), row)


# ---- storage / coalesce / transpose KATs ----------------------------------
def test_storage_sort_kat(kats):
    k = kats["storage_sort"]
    st = so.Storage(k["row"], k["col"], np.array(k["value"], np.float32))
    assert st.row.tolist() == k["out_row"] and st.col.tolist() == k["out_col"]
    assert st.value.tolist() == k["out_value"]
    assert [st.M, st.N] == k["sparse_sizes"]


def test_storage_caching_kat(kats):
    k = kats["storage_caching"]
    st = so.Storage(k["row"], k["col"])
    assert st.rowcount().tolist() == k["rowcount"]
    assert st.rowptr().tolist() == k["rowptr"]
    assert st.colcount().tolist() == k["colcount"]
    assert st.colptr().tolist() == k["colptr"]
    assert st.csr2csc().tolist() == k["csr2csc"]
    assert st.csc2csr().tolist() == k["csc2csr"]
    assert np.array_equal(st.rowptr(), oracle.ind2ptr(st.row, st.M))


def test_storage_set_value_csc_kat(kats):
    k = kats["storage_set_value_csc"]
    st = so.Storage(k["row"], k["col"], np.array(k["value"], np.float32))
    # storage.py:246-247: value given in CSC order -> value[csc2csr]
    assert np.array(k["value"])[st.csc2csr()].tolist() == k["csc_value"]


def test_storage_coalesce_kat(kats):
    k = kats["storage_coalesce"]
    st = so.Storage(k["row"], k["col"], np.array(k["value"], np.float32))
    assert not st.is_coalesced()
    st = st.coalesce()
    assert st.is_coalesced()
    assert st.row.tolist() == k["out_row"] and st.col.tolist() == k["out_col"]
    assert st.value.tolist() == k["out_value"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_coalesce_kats(kats, dtype):
    k = kats["coalesce"]
    index = np.array([k["row"], k["col"]])
    idx, val = so.coalesce(index, None, k["m"], k["n"])
    assert idx.tolist() == k["out_index"] and val is None
    idx, val = so.coalesce(index, np.array(k["value"], dtype), k["m"], k["n"])
    assert idx.tolist() == k["out_index"] and val.tolist() == k["out_add"]
    idx, val = so.coalesce(index, np.array(k["value"], dtype), k["m"], k["n"], op="max")
    assert idx.tolist() == k["out_index"] and val.tolist() == k["out_max"]


def test_transpose_kats(kats):
    for name in ("transpose_matrix", "transpose"):
        k = kats[name]
        idx, val = so.transpose(np.array([k["row"], k["col"]]), np.array(k["value"], np.float32), k["m"], k["n"])
        assert idx.tolist() == k["out_index"]
        assert val.tolist() == k["out_value"]


def test_reduce_dim_none_kat(kats):
    k = kats["reduce_dim_none"]
    st = so.Storage(k["row"], k["col"], np.array(k["value"], np.float32))
    for r in ("sum", "mean", "max", "min"):
        assert so.reduction(st, None, r) == k[r]


def test_to_symmetric_kat(kats):
    k = kats["to_symmetric"]
    st = so.to_symmetric(so.Storage(k["row"], k["col"], np.array(k["value"], np.int64)))
    assert so.dense(st).tolist() == k["dense"]
    assert np.array_equal(so.dense(st), so.dense(st).T)


def test_add_mul_kats(kats):
    k = kats["add_mul_sparse_sparse"]
    A = so.Storage(k["rowA"], k["colA"], np.array(k["valueA"], np.float32))
    B = so.Storage(k["rowB"], k["colB"], np.array(k["valueB"], np.float32))
    C = so.add(A, B)
    assert C.row.tolist() == k["add_row"] and C.col.tolist() == k["add_col"]
    assert C.value.tolist() == k["add_value"]
    C = so.mul(A, B)
    assert C.row.tolist() == k["mul_row"] and C.col.tolist() == k["mul_col"]
    assert C.value.tolist() == k["mul_value"]
    e = k["mul_empty"]
    C = so.mul(so.Storage(e["rowA"], e["colA"], np.array(e["valueA"], np.float32)),
               so.Storage(e["rowB"], e["colB"], np.array(e["valueB"], np.float32)))
    assert C.row.size == 0 and C.col.size == 0 and C.value.size == 0


def test_eye_caches_kat(kats):
    """test/test_eye.py:42-66: the caches an identity must carry equal the ones
    the generic cache fill computes from its (row, col)."""
    for c in kats["eye_caches"]["cases"]:
        st = so.Storage(c["row"], c["col"], None, (c["M"], c["N"]), is_sorted=True)
        for name in ("rowptr", "rowcount", "colptr", "colcount", "csr2csc", "csc2csr"):
            assert getattr(st, name)().tolist() == c[name], (c["M"], c["N"], name)


@pytest.mark.parametrize("threads", [1, 4])
def test_c_coalesce_kats_and_numpy(kats, threads):
    """coalesce_oracle.c (the timed CPU baseline of the sort/coalesce rows)
    gives the reference's known answers and the numpy restatement's results."""
    k = kats["coalesce"]
    val = np.array(k["value"], np.float32)
    for op, key in (("add", "out_add"), ("max", "out_max")):
        idx, out = oracle.coalesce_c(k["row"], k["col"], val, k["m"], k["n"], op, threads)
        assert idx.tolist() == k["out_index"] and out.tolist() == k[key]
    idx, out = oracle.coalesce_c(k["row"], k["col"], None, k["m"], k["n"], "add", threads)
    assert idx.tolist() == k["out_index"] and out is None
    k = kats["storage_coalesce"]
    idx, out = oracle.coalesce_c(k["row"], k["col"], np.array(k["value"], np.float32), 2, 2, "add", threads)
    assert idx.tolist() == [k["out_row"], k["out_col"]] and out.tolist() == k["out_value"]
    idx, out = oracle.coalesce_c([], [], np.zeros(0, np.float32), 4, 4, "add", threads)
    assert idx.shape == (2, 0) and out.shape == (0,)
    rng = np.random.default_rng(7)
    for M, N, nnz, D in ((1000, 1000, 10_000, None), (300, 70_000, 200_000, 2), (5, 3, 4000, 3)):
        row, col = rng.integers(0, M, nnz), rng.integers(0, N, nnz)
        v = rng.standard_normal((nnz,) if D is None else (nnz, D)).astype(np.float32)
        for op in ("add", "mean", "min", "max"):
            ref_idx, ref_v = so.coalesce(np.stack([row, col]), v, M, N, op)
            idx, out = oracle.coalesce_c(row, col, v, M, N, op, threads)
            assert np.array_equal(idx, ref_idx)
            if op in ("min", "max"):
                assert np.array_equal(out, ref_v)
            else:  # the numpy side switches to reduceat (pairwise order) above 4096 values
                np.testing.assert_allclose(out, ref_v, rtol=1e-5, atol=1e-6)
        # already-sorted input skips the sort (storage.py:163)
        idx2, out2 = oracle.coalesce_c(idx[0], idx[1], out, M, N, "add", threads)
        assert np.array_equal(idx2, idx) and np.array_equal(out2, out)


@pytest.mark.parametrize("threads", [1, 3])
def test_c_index_sort_is_the_stable_permutation(threads):
    rng = np.random.default_rng(8)
    for n, mx in ((0, 10), (1, 1), (1000, 7), (50_000, 1 << 20), (20_000, 1 << 47)):
        keys = rng.integers(0, mx, n)
        s, p = oracle.index_sort_c(keys, mx, threads)
        assert np.array_equal(p, so.index_sort(keys))
        assert np.array_equal(s, keys[p])


def test_spspmm_readme_kat_and_scipy(kats):
    k = kats["spspmm"]
    idx, val = oracle.spspmm(k["indexA"], k["valueA"], k["indexB"], k["valueB"], k["m"], k["k"], k["n"])
    assert idx.tolist() == k["indexC"] and val.tolist() == k["valueC"]
    rng = np.random.default_rng(9)
    for m, kk, n, nnzA, nnzB in ((50, 40, 30, 300, 200), (400, 300, 500, 5000, 4000), (7, 5000, 9, 900, 2000)):
        keyA, keyB = np.unique(rng.integers(0, m * kk, nnzA)), np.unique(rng.integers(0, kk * n, nnzB))
        iA, iB = np.stack([keyA // kk, keyA % kk]), np.stack([keyB // n, keyB % n])
        # small integers: every fp32 sum is exact, so scipy's order cannot matter
        vA = rng.integers(-4, 5, keyA.size).astype(np.float32)
        vB = rng.integers(-4, 5, keyB.size).astype(np.float32)
        idx, val = oracle.spspmm(iA, vA, iB, vB, m, kk, n)
        A = scipy.sparse.csr_matrix((vA, (iA[0], iA[1])), (m, kk))
        B = scipy.sparse.csr_matrix((vB, (iB[0], iB[1])), (kk, n))
        dense = np.zeros((m, n), np.float32)
        dense[idx[0], idx[1]] = val
        assert np.array_equal(dense, (A @ B).toarray())
        # structure: every (i, j) with a structural product, explicit zeros kept, row-major sorted
        S = (abs(A).sign() @ abs(B).sign()).tocoo()
        structural = np.unique(np.asarray(S.row, np.int64) * n + S.col)
        A1 = scipy.sparse.csr_matrix((np.ones(keyA.size), (iA[0], iA[1])), (m, kk))
        B1 = scipy.sparse.csr_matrix((np.ones(keyB.size), (iB[0], iB[1])), (kk, n))
        full = (A1 @ B1).tocoo()
        assert np.array_equal(idx[0] * n + idx[1], np.unique(np.asarray(full.row, np.int64) * n + full.col))
        assert structural.size <= idx.shape[1]


def test_sample_adj_kat_and_properties(kats):
    k = kats["sample_adj"]
    M = k["sparse_sizes"][0]
    rowptr = oracle.ind2ptr(k["row"], M)
    col, subset = np.array(k["col"]), np.array(k["subset"])
    value = np.arange(len(k["row"]))
    rp, c, n_id, e_id = oracle.sample_adj(rowptr, col, subset, -1)
    a = k["all_neighbors"]
    assert n_id.tolist() == a["n_id"]
    assert oracle.ptr2ind(rp, c.size).tolist() == a["row"] and c.tolist() == a["col"]
    assert value[e_id].tolist() == a["val"]
    for seed in range(5):
        rp, c, n_id, e_id = oracle.sample_adj(rowptr, col, subset, 2, True, seed)
        assert c.size == k["nnz_2_with_replacement"]
        rp, c, n_id, e_id = oracle.sample_adj(rowptr, col, subset, 2, False, seed)
        assert c.size == k["nnz_2_without_replacement"]
    # larger graph: picks are edges of the right row, distinct without replacement, ids consistent
    rng = np.random.default_rng(3)
    row, rowptr, col, _ = random_csr(3000, 3000, 40_000, 3)
    subset = rng.permutation(3000)[:500]
    for kk, rep in ((5, False), (5, True), (50, False), (-1, False)):
        rp, c, n_id, e_id = oracle.sample_adj(rowptr, col, subset, kk, rep, seed=11)
        assert np.array_equal(n_id[:500], subset) and np.unique(n_id).size == n_id.size
        assert np.array_equal(n_id[c], col[e_id])
        r = oracle.ptr2ind(rp, c.size)
        assert np.array_equal(row[e_id], subset[r])
        deg = rowptr[subset + 1] - rowptr[subset]
        want = deg if kk < 0 else (np.where(deg > 0, kk, 0) if rep else np.minimum(deg, kk))
        assert np.array_equal(rp[1:] - rp[:-1], want)
        assert np.all((c[1:] >= c[:-1]) | (r[1:] != r[:-1]))
        if not rep:
            assert np.unique(e_id).size == e_id.size
        # a different seed gives a different sample, the same seed the same one
        if kk == 5:
            again = oracle.sample_adj(rowptr, col, subset, kk, rep, seed=11)
            other = oracle.sample_adj(rowptr, col, subset, kk, rep, seed=12)
            assert np.array_equal(again[3], e_id) and not np.array_equal(other[3], e_id)


def test_segment_csr_fast_matches_loop():
    rng = np.random.default_rng(0)
    src = rng.integers(-50, 50, (300, 3)).astype(np.int64)
    cuts = np.sort(rng.integers(0, 301, 40))
    indptr = np.concatenate([[0], cuts, [300]]).astype(np.int64)
    for r in ("sum", "mean", "min", "max"):
        assert np.array_equal(so.segment_csr(src, indptr, r), so.segment_csr_fast(src, indptr, r))


def test_coalesce_random_vs_scipy():
    """C1-shaped: 10k-edge unsorted COO with duplicates; scipy sums duplicates."""
    rng = np.random.default_rng(0)
    M = N = 1000
    row, col = rng.integers(0, M, 10000), rng.integers(0, N, 10000)
    val = rng.integers(-8, 8, 10000).astype(np.float32)  # exact in fp32
    idx, out = so.coalesce(np.stack([row, col]), val, M, N)
    ref = scipy.sparse.coo_matrix((val, (row, col)), (M, N)).tocsr()
    ref.sum_duplicates()
    ref.sort_indices()
    refc = ref.tocoo()
    # scipy drops nothing here (explicit zeros are kept by sum_duplicates)
    assert np.array_equal(idx[0], refc.row) and np.array_equal(idx[1], refc.col)
    assert np.array_equal(out, refc.data)


# ---- SpMM: README KAT + third-party cross-checks ---------------------------
def test_spmm_readme_kat(kats):
    k = kats["spmm"]
    row, col = np.array(k["index"])
    rowptr = oracle.ind2ptr(row, k["m"])
    out, _ = oracle.spmm("sum", rowptr, col, np.array(k["value"], np.float32), np.array(k["matrix"], np.float32))
    assert out.tolist() == k["out"]


def test_spmm_survey_vectors():
    """torch-CPU outputs on the README matrix + one empty row."""
    rowptr = np.array([0, 2, 3, 5, 5])
    col = np.array([0, 2, 1, 0, 1])
    val = np.array([1, 2, 4, 1, 3], np.float32)
    B = np.array([[1, 4], [2, 5], [3, 6]], np.float32)
    assert oracle.spmm("sum", rowptr, col, val, B)[0].tolist() == [[7, 16], [8, 20], [7, 19], [0, 0]]
    assert oracle.spmm("mean", rowptr, col, val, B)[0].tolist() == [[3.5, 8], [8, 20], [3.5, 9.5], [0, 0]]
    out, arg = oracle.spmm("max", rowptr, col, val, B)
    assert out.tolist() == [[6, 12], [8, 20], [6, 15], [0, 0]]
    assert arg.tolist() == [[1, 1], [2, 2], [4, 4], [5, 5]]
    out, arg = oracle.spmm("min", rowptr, col, val, B)
    assert out.tolist() == [[1, 4], [8, 20], [1, 4], [0, 0]]
    assert arg.tolist() == [[0, 0], [2, 2], [3, 3], [5, 5]]
    row = oracle.ptr2ind(rowptr, 5)
    g = np.ones((4, 2), np.float32)
    assert oracle.spmm_mat_bw("sum", row, rowptr, col, val, g, 3).tolist() == [[2, 2], [7, 7], [2, 2]]
    assert oracle.spmm_value_bw("sum", row, rowptr, col, B, g).tolist() == [5, 9, 7, 5, 7]


def _torch_csr(rowptr, col, val, M, N, requires_grad=False):
    v = torch.tensor(val, dtype=torch.float64, requires_grad=requires_grad)
    return torch.sparse_csr_tensor(torch.tensor(rowptr), torch.tensor(col), v, size=(M, N)), v


@pytest.mark.parametrize("reduce,treduce", [("sum", "sum"), ("mean", "mean"), ("max", "amax"), ("min", "amin")])
@pytest.mark.parametrize("K", [1, 7, 32])
def test_spmm_vs_torch_cpu(reduce, treduce, K):
    M, N, nnz = 200, 150, 1500
    row, rowptr, col, val = random_csr(M, N, nnz, seed=K, sort_cols=True)
    # torch's CSR reducers need coalesced columns per row: dedup (row, col)
    key = row * N + col
    keep = np.concatenate([[True], key[1:] != key[:-1]])
    row, col, val = row[keep], col[keep], val[keep]
    rowptr = oracle.ind2ptr(row, M)
    B = np.random.default_rng(1).standard_normal((N, K)).astype(np.float32)
    out, _ = oracle.spmm(reduce, rowptr, col, val, B)
    A, _ = _torch_csr(rowptr, col, val, M, N)
    ref = torch.sparse.mm(A, torch.tensor(B, dtype=torch.float64), treduce).numpy()
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    assert np.all(np.abs(out - ref) <= 1e-5 * S + 1e-30)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_spmm_backward_vs_torch_cpu(reduce):
    M, N, nnz, K = 120, 90, 800, 5
    row, rowptr, col, val = random_csr(M, N, nnz, seed=11, sort_cols=True)
    key = row * N + col
    keep = np.concatenate([[True], key[1:] != key[:-1]])
    row, col, val = row[keep], col[keep], val[keep]
    rowptr = oracle.ind2ptr(row, M)
    rng = np.random.default_rng(2)
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    A, v = _torch_csr(rowptr, col, val, M, N, requires_grad=True)
    Bt = torch.tensor(B, dtype=torch.float64, requires_grad=True)
    torch.sparse.mm(A, Bt, reduce).backward(torch.tensor(G, dtype=torch.float64))
    gB = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N)
    gV = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
    np.testing.assert_allclose(gB, Bt.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gV, v.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_spmm_vs_scipy_with_duplicates_and_skew():
    row, rowptr, col, val = skewed_csr(300, 200, seed=5, long_rows=(0, 17), long_deg=400)
    B = np.random.default_rng(3).standard_normal((200, 16)).astype(np.float32)
    out, _ = oracle.spmm("sum", rowptr, col, val, B)
    ref = scipy.sparse.csr_matrix((val.astype(np.float64), col, rowptr), (300, 200)) @ B.astype(np.float64)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    assert np.all(np.abs(out - ref) <= 1e-5 * S + 1e-30)
    # OpenMP entry point = same arithmetic per row
    out2, _ = oracle.spmm("sum", rowptr, col, val, B, threads=4)
    assert np.array_equal(out, out2)


def test_spmm_minmax_backward_matches_definition():
    row, rowptr, col, val = random_csr(50, 40, 300, seed=9)
    rng = np.random.default_rng(4)
    B = rng.standard_normal((40, 6)).astype(np.float32)
    G = rng.standard_normal((50, 6)).astype(np.float32)
    out, arg = oracle.spmm("max", rowptr, col, val, B)
    gv, gm = oracle.spmm_minmax_bw(col, val, B, G, arg)
    gv_ref = np.zeros(300, np.float64)
    gm_ref = np.zeros((40, 6), np.float64)
    for i in range(50):
        for k in range(6):
            e = arg[i, k]
            if e == 300:
                continue
            gv_ref[e] += float(B[col[e], k]) * float(G[i, k])
            gm_ref[col[e], k] += float(val[e]) * float(G[i, k])
    np.testing.assert_allclose(gv, gv_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gm, gm_ref, rtol=1e-5, atol=1e-5)


# ---- third-party golden vectors (tests/golden/make_golden.py) -----------------------------------
# torch-CPU torch.sparse.mm (forward and autograd), numpy ufunc.at, torch.segment_reduce: frozen
# answers for the rows no reference fixture covers.  The oracle is held to them here, the HIP
# path in tests/test_golden_gpu.py.

@pytest.fixture(scope="module")
def golden():
    return np.load(ROOT / "tests" / "golden" / "third_party.npz")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_oracle_spmm_forward_backward_vs_golden(golden, tag, reduce):
    g = golden
    rowptr, col, val, B, G = (g[f"spmm_{tag}_{k}"] for k in ("rowptr", "col", "val", "B", "G"))
    M, N = rowptr.size - 1, B.shape[0]
    row = np.repeat(np.arange(M, dtype=np.int64), np.diff(rowptr))
    out, arg = oracle.spmm(reduce, rowptr, col, val, B)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    assert np.all(np.abs(out - g[f"spmm_{tag}_{reduce}_out"]) <= 1e-5 * S + 1e-30)
    if reduce in ("sum", "mean"):
        gm = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N)
        gv = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
        sm = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(G), N)
        sv = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(B), np.abs(G))
    else:
        gv, gm = oracle.spmm_minmax_bw(col, val, B, G, arg)
        sv, sm = oracle.spmm_minmax_bw(col, np.abs(val), np.abs(B), np.abs(G), arg)
    assert np.all(np.abs(gm - g[f"spmm_{tag}_{reduce}_gmat"]) <= 1e-5 * sm + 1e-30)
    assert np.all(np.abs(gv - g[f"spmm_{tag}_{reduce}_gval"]) <= 1e-5 * sv + 1e-30)


def test_oracle_reduce_and_coalesce_vs_golden(golden):
    g = golden
    M, N = (int(x) for x in g["red_shape"])
    st = so.Storage(g["red_row"], g["red_col"], g["red_val"], (M, N))
    for dim in (0, 1):
        for reduce in ("sum", "mean", "min", "max"):
            ref = g[f"red_dim{dim}_{reduce}"]
            got = so.reduction(st, dim, reduce)
            if reduce in ("min", "max"):
                assert np.array_equal(got, ref), (dim, reduce)
            else:
                np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    m, n = (int(x) for x in g["co_shape"])
    index = np.stack([g["co_row"], g["co_col"]])
    for op in ("add", "mean", "min", "max"):
        gi, gv = so.coalesce(index, g["co_val"], m, n, op)
        assert np.array_equal(gi, g["co_index"])
        if op in ("min", "max"):
            assert np.array_equal(gv, g[f"co_{op}"])
        else:
            np.testing.assert_allclose(gv, g[f"co_{op}"], rtol=1e-5, atol=1e-6)
        ci, cv = oracle.coalesce_c(g["co_row"], g["co_col"], g["co_val"], m, n, op)[:2]
        assert np.array_equal(ci, g["co_index"])
        np.testing.assert_allclose(cv, g[f"co_{op}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_spspmm_vs_golden(golden, tag):
    """scipy.sparse CSR @ CSR on seeded matrices (tests/golden/make_golden.py): the structural product —
    entries whose terms cancel stay stored (tag c holds 22 of them) — and its values to 1e-5 * (|A| @ |B|).
    The reference holds one 3 x 3 vector for spspmm (README.md:308-353)."""
    g = golden
    m, k, n = (int(x) for x in g[f"spspmm_{tag}_shape"])
    index, value = oracle.spspmm(g[f"spspmm_{tag}_indexA"], g[f"spspmm_{tag}_valueA"], g[f"spspmm_{tag}_indexB"],
                                 g[f"spspmm_{tag}_valueB"], m, k, n)
    assert np.array_equal(index, g[f"spspmm_{tag}_index"])
    assert np.all(np.abs(value - g[f"spspmm_{tag}_value"]) <= 1e-5 * g[f"spspmm_{tag}_abs"] + 1e-30)
    if tag == "c":  # integer-valued operands: exact, cancelled entries are stored zeros
        assert np.array_equal(value, g[f"spspmm_{tag}_value"]) and int((value == 0).sum()) == int(g["spspmm_c_cancelled"][0]) > 0
