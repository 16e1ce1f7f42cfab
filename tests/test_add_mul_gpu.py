"""GPU: add / mul on the SparseTensor surface against the
reference's known answers (test/test_add.py, test/test_mul.py) and scipy on
seeded inputs."""
import numpy as np
import pytest
import scipy.sparse
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.float32, torch.float64, torch.int32, torch.int64, torch.bfloat16]


def idx(x):
    return torch.tensor(x, dtype=torch.int64, device="cuda")


def make(k, which, dtype):
    from paddle_sparse_amd import SparseTensor

    return SparseTensor(row=idx(k["row" + which]), col=idx(k["col" + which]),
                        value=torch.tensor(k["value" + which], dtype=dtype, device="cuda"))


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_kat(kats, dtype):
    k = kats["add_mul_sparse_sparse"]
    C = make(k, "A", dtype) + make(k, "B", dtype)
    row, col, value = C.coo()
    assert row.tolist() == k["add_row"] and col.tolist() == k["add_col"]
    assert torch.equal(value, torch.tensor(k["add_value"], dtype=dtype, device="cuda"))


@pytest.mark.parametrize("dtype", DTYPES)
def test_sparse_sparse_mul_kat(kats, dtype):
    k = kats["add_mul_sparse_sparse"]
    C = make(k, "A", dtype) * make(k, "B", dtype)
    row, col, value = C.coo()
    assert row.tolist() == k["mul_row"] and col.tolist() == k["mul_col"]
    assert torch.equal(value, torch.tensor(k["mul_value"], dtype=dtype, device="cuda"))
    e = k["mul_empty"]
    C = make(e, "A", dtype) * make(e, "B", dtype)
    row, col, value = C.coo()
    assert row.tolist() == [] and col.tolist() == [] and value.tolist() == []


def _random(M, N, nnz, seed):
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(seed)
    key = np.unique(rng.integers(0, M * N, nnz))
    row, col = key // N, key % N
    val = rng.integers(-5, 6, key.size).astype(np.float32)
    val[val == 0] = 1
    t = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N))
    return t, scipy.sparse.csr_matrix((val, (row, col)), (M, N))


def test_add_mul_random_vs_scipy():
    A, sa = _random(300, 200, 5000, 1)
    B, sb = _random(300, 200, 4000, 2)
    assert np.array_equal((A + B).to_dense().cpu().numpy(), (sa + sb).toarray())
    C = A * B
    assert np.array_equal(C.to_dense().cpu().numpy(), sa.multiply(sb).toarray())
    assert C.nnz() == sa.multiply(sb).nnz
    # operands of different shapes (add.py:41-43: sizes are the elementwise max)
    D, sd = _random(350, 150, 3000, 3)
    S = A + D
    assert S.sparse_sizes() == (350, 200)
    ref = np.zeros((350, 200), np.float32)
    ref[:300, :200] += sa.toarray()
    ref[:350, :150] += sd.toarray()
    assert np.array_equal(S.to_dense().cpu().numpy(), ref)


def test_add_mul_symmetric_vs_oracle():
    """Bit-exact (row, col, value) against oracle.storage_oracle on seeded inputs."""
    from oracle import storage_oracle as so

    def to_oracle(t):
        row, col, value = t.coo()
        return so.Storage(row.cpu().numpy(), col.cpu().numpy(), value.cpu().numpy(),
                          t.sparse_sizes(), is_sorted=True)

    def same(t, o):
        row, col, value = t.coo()
        assert np.array_equal(row.cpu().numpy(), o.row) and np.array_equal(col.cpu().numpy(), o.col)
        assert np.array_equal(value.cpu().numpy(), o.value)
        assert t.sparse_sizes() == (o.M, o.N)

    A, _ = _random(2000, 1500, 60000, 11)
    B, _ = _random(2000, 1500, 50000, 12)
    same(A + B, so.add(to_oracle(A), to_oracle(B)))
    same(A * B, so.mul(to_oracle(A), to_oracle(B)))
    for reduce in ("sum", "max", "min"):
        same(A.to_symmetric(reduce), so.to_symmetric(to_oracle(A), reduce))


@pytest.mark.parametrize("na,nb,span,seed", [(0, 0, 10, 0), (0, 5, 10, 1), (7, 0, 10, 2), (1, 1, 1, 3), (5000, 3000, 50, 4),
                                             (100000, 250000, 1 << 40, 5), (300000, 300000, 1000, 6), (2049, 4097, 1 << 20, 7)])
def test_merge_sorted_is_the_stable_sort_of_the_concatenation(na, nb, span, seed):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(seed)
    a = np.sort(rng.integers(0, span, na, dtype=np.int64))
    b = np.sort(rng.integers(0, span, nb, dtype=np.int64))
    pa = torch.from_numpy(rng.standard_normal(na).astype(np.float32)).cuda()
    pb = torch.from_numpy(rng.standard_normal(nb).astype(np.float32)).cuda()
    cat = np.concatenate([a, b])
    ref = np.argsort(cat, kind="stable")
    merged, source, pay = ops.merge_sorted(idx(a), idx(b), pa, pb)
    assert np.array_equal(merged.cpu().numpy(), cat[ref])
    assert np.array_equal(source.cpu().numpy(), ref)
    assert torch.equal(pay.cpu(), torch.cat([pa, pb]).cpu()[torch.from_numpy(ref)])
    merged, source, pay = ops.merge_sorted(idx(a), idx(b), want_source=False)
    assert source is None and pay is None and np.array_equal(merged.cpu().numpy(), cat[ref])


def test_merge_sorted_one_sided_and_unsorted_inputs():
    """All of one stream before / after the other (tiles made of one stream
    only), long runs of one key, and unsorted input: order unspecified then,
    but the call must stay inside its buffers and return."""
    from paddle_sparse_amd import ops

    lo, hi = np.arange(0, 50000, dtype=np.int64), np.arange(10 ** 6, 10 ** 6 + 70001, dtype=np.int64)
    for a, b in ((lo, hi), (hi, lo), (np.full(30000, 7, np.int64), np.full(41000, 7, np.int64)),
                 (np.repeat(np.arange(10, dtype=np.int64), 5000), np.repeat(np.arange(5, 15, dtype=np.int64), 3000))):
        cat = np.concatenate([a, b])
        ref = np.argsort(cat, kind="stable")
        merged, source, _ = ops.merge_sorted(idx(a), idx(b))
        assert np.array_equal(merged.cpu().numpy(), cat[ref]) and np.array_equal(source.cpu().numpy(), ref)
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 1 << 40, 100000, dtype=np.int64), rng.integers(0, 1 << 40, 77777, dtype=np.int64)
    pa, pb = torch.ones(a.size, device="cuda"), torch.ones(b.size, device="cuda")
    merged, source, pay = ops.merge_sorted(idx(a), idx(b), pa, pb)
    torch.cuda.synchronize()
    assert merged.numel() == a.size + b.size and bool(((pay == 0) | (pay == 1)).all())


@pytest.mark.parametrize("dtype,tail", [(torch.float64, ()), (torch.float32, (3,)), (torch.int64, ()), (torch.float16, ())])
def test_add_mul_symmetric_merge_path_other_value_types(dtype, tail):
    """Values that cannot ride the merge as a 4-byte payload go through the
    source index; same results as the numpy oracle, bit for bit."""
    from oracle import storage_oracle as so
    from paddle_sparse_amd import SparseTensor

    def make(seed, nnz):
        rng = np.random.default_rng(seed)
        key = np.unique(rng.integers(0, 3000 * 2500, nnz))
        val = rng.integers(-4, 5, (key.size,) + tail).astype(np.float64)
        val[val == 0] = 1
        v = torch.from_numpy(val).to(dtype)
        t = SparseTensor(row=idx(key // 2500), col=idx(key % 2500), value=v.cuda(), sparse_sizes=(3000, 2500))
        return t, so.Storage(key // 2500, key % 2500, v.numpy() if dtype != torch.float16 else v.float().numpy(),
                             (3000, 2500), is_sorted=True)

    def same(t, o):
        row, col, value = t.coo()
        assert np.array_equal(row.cpu().numpy(), o.row) and np.array_equal(col.cpu().numpy(), o.col)
        assert np.array_equal(value.cpu().double().numpy(), np.asarray(o.value, dtype=np.float64))

    (A, oa), (B, ob) = make(21, 90000), make(22, 70000)
    same(A + B, so.add(oa, ob))
    if not tail:
        same(A * B, so.mul(oa, ob))
    for reduce in ("sum", "max"):
        same(A.to_symmetric(reduce), so.to_symmetric(oa, reduce))
    # a value-less operand: the union has no values (add.py:37-39)
    C = A + B.set_value(None, layout="coo")
    assert not C.has_value() and C.nnz() == so.add(oa, ob).row.size


def test_to_symmetric_kat(kats):
    from paddle_sparse_amd import SparseTensor

    k = kats["to_symmetric"]
    for dtype in DTYPES:
        t = SparseTensor(row=idx(k["row"]), col=idx(k["col"]),
                         value=torch.tensor(k["value"], dtype=dtype, device="cuda")).to_symmetric()
        assert t.to_dense().cpu().to(torch.int64).tolist() == k["dense"]
        assert t.is_symmetric()


def test_dense_broadcast_add_mul():
    A, sa = _random(60, 40, 500, 4)
    dense = sa.toarray()
    mask = dense != 0
    r = torch.arange(60, dtype=torch.float32, device="cuda").view(-1, 1)
    c = torch.arange(40, dtype=torch.float32, device="cuda").view(1, -1)
    for op, npop in (("add", np.add), ("mul", np.multiply)):
        got = getattr(A, op)(r).to_dense().cpu().numpy()
        assert np.array_equal(got, np.where(mask, npop(dense, r.cpu().numpy()), 0))
        got = getattr(A, op)(c).to_dense().cpu().numpy()
        assert np.array_equal(got, np.where(mask, npop(dense, c.cpu().numpy()), 0))
    with pytest.raises(ValueError, match="Size mismatch"):
        A + torch.ones(3, 3, device="cuda")
    # nnz-wise and in-place forms
    v = torch.full((A.nnz(),), 2.0, device="cuda")
    assert np.array_equal(A.mul_nnz(v, layout="coo").to_dense().cpu().numpy(), dense * 2)
    assert np.array_equal(A.add_nnz(v, layout="coo").to_dense().cpu().numpy(), np.where(mask, dense + 2, 0))
    B = A.copy()
    B.storage._value = A.storage.value().clone()
    B *= c
    assert np.array_equal(B.to_dense().cpu().numpy(), np.where(mask, dense * c.cpu().numpy(), 0))


def test_mul_requires_coalesced_operands():
    from paddle_sparse_amd import SparseTensor

    dup = SparseTensor(row=idx([0, 0]), col=idx([1, 1]), value=torch.ones(2, device="cuda"))
    ok = SparseTensor(row=idx([0]), col=idx([1]), value=torch.ones(1, device="cuda"))
    with pytest.raises(ValueError, match="not coalesced"):
        dup * ok


def test_overload():
    """test/test_overload.py:6-21: both operand orders, column and row vectors."""
    from paddle_sparse_amd import SparseTensor

    mat = SparseTensor(row=idx([0, 1, 1, 2, 2]), col=idx([1, 0, 2, 1, 2]))
    dense = mat.to_dense(dtype=torch.int64)
    for other in (torch.tensor([1, 2, 3], device="cuda").view(3, 1), torch.tensor([1, 2, 3], device="cuda").view(1, 3)):
        for got in (other + mat, mat + other):
            assert torch.equal(got.to_dense(), torch.where(dense != 0, dense + other, 0))
        for got in (other * mat, mat * other):
            assert torch.equal(got.to_dense(), dense * other)
