"""GPU: seeded random shapes through the public surface against the oracle —
odd sizes, empty rows/columns, hubs, every K tile class and reduce — to catch
what the hand-picked cases of the other files do not (kernel selection happens
inside the package, so this walks whichever kernel a shape lands on)."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import storage_oracle as so

pytestmark = pytest.mark.gpu

# PSA_FUZZ=10 runs ten times as many seeds (a soak run; the default keeps the suite short)
SCALE = int(os.environ.get("PSA_FUZZ", "1"))


def idx(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.int64).cuda()


def random_graph(rng):
    M, N = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    style = rng.integers(0, 3)
    if style == 0:  # uniform
        nnz = int(rng.integers(0, 20_000))
        row, col = rng.integers(0, M, nnz), rng.integers(0, N, nnz)
    elif style == 1:  # a few hub rows and hub columns
        nnz = int(rng.integers(1, 30_000))
        row = np.where(rng.random(nnz) < 0.3, rng.integers(0, min(M, 3), nnz), rng.integers(0, M, nnz))
        col = np.where(rng.random(nnz) < 0.3, rng.integers(0, min(N, 2), nnz), rng.integers(0, N, nnz))
    else:  # very sparse: most rows empty
        nnz = int(rng.integers(0, max(2, M // 4)))
        row, col = rng.integers(0, M, nnz), rng.integers(0, N, nnz)
    key = np.unique(row.astype(np.int64) * N + col)  # coalesced, row-major
    return M, N, key // N, key % N


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_spmm_forward_backward_random_shapes(seed):
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(1000 + seed)
    M, N, row, col = random_graph(rng)
    K = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 20, 31, 32, 33, 64, 65, 100, 128, 130, 160, 192, 224, 256, 260, 300, 320, 384, 512]))
    reduce = ["sum", "mean", "min", "max"][seed % 4]
    has_value = bool(rng.integers(0, 2))
    nnz = row.size
    val = rng.standard_normal(nnz).astype(np.float32) if has_value else None
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)

    v = torch.from_numpy(val).cuda().requires_grad_() if has_value else None
    Bt = torch.from_numpy(B).cuda().requires_grad_()
    a = SparseTensor(row=idx(row), col=idx(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    out = a.matmul(Bt, reduce)
    out.backward(torch.from_numpy(G).cuda())

    ref, arg = oracle.spmm(reduce, rowptr, col, val, B)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    assert np.all(np.abs(out.detach().cpu().numpy() - ref) <= 1e-5 * S + 1e-30), (M, N, K, reduce)
    if reduce in ("sum", "mean"):
        gB = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N)
        gV = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
    else:
        gV, gB = oracle.spmm_minmax_bw(col, val, B, G, arg)
    np.testing.assert_allclose(Bt.grad.cpu().numpy(), gB, rtol=2e-4, atol=2e-4, err_msg=str((M, N, K, reduce)))
    if has_value:
        np.testing.assert_allclose(v.grad.cpu().numpy(), gV, rtol=2e-4, atol=2e-4, err_msg=str((M, N, K, reduce)))


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_spmm_fixed_adjacency_random_shapes(seed):
    """Fixed edge weights (gradient wrt the dense operand only), fp32 and bf16 operands, with the
    hub-row plans switched on for small matrices: the per-matrix choices of matmul.py — edge ranges,
    hub-row copies, the masked edge-range backward, the half-width backward — on whatever shape comes."""
    import paddle_sparse_amd.storage as st_mod
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(3000 + seed)
    M, N, row, col = random_graph(rng)
    K = int(rng.choice([4, 8, 16, 32, 64, 72, 128, 136, 256, 264]))
    reduce = ["sum", "mean", "min", "max"][seed % 4]
    half = seed % 3 == 0 and reduce in ("sum", "mean")
    has_value = bool(rng.integers(0, 2))
    nnz = row.size
    val = rng.standard_normal(nnz).astype(np.float32) if has_value else None
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    if half:  # the oracle sees what the kernels see: bf16-rounded operands
        B = torch.from_numpy(B).to(torch.bfloat16).float().numpy()
        G = torch.from_numpy(G).to(torch.bfloat16).float().numpy()
    rowptr = oracle.ind2ptr(row, M)
    dt = torch.bfloat16 if half else torch.float32
    Bt = torch.from_numpy(B).cuda().to(dt).requires_grad_()
    a = SparseTensor(row=idx(row), col=idx(col), value=None if val is None else torch.from_numpy(val).cuda(),
                     sparse_sizes=(M, N), is_sorted=True)
    old = st_mod.HOT_COLUMNS
    st_mod.HOT_COLUMNS = 64
    try:
        out = a.matmul(Bt, reduce)
        out.backward(torch.from_numpy(G).cuda().to(dt))
    finally:
        st_mod.HOT_COLUMNS = old
    ref, arg = oracle.spmm(reduce, rowptr, col, val, B)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    eps = 2.0 ** -8 if half else 0.0
    assert np.all(np.abs(out.detach().float().cpu().numpy() - ref) <= 1e-5 * S + eps * np.abs(ref) + 1e-30), (M, N, K, reduce)
    ones = np.ones(nnz, np.float32)
    if reduce in ("sum", "mean"):
        gB = oracle.spmm_mat_bw(reduce, row, rowptr, col, ones if val is None else val, G, N)
        scale = oracle.spmm_mat_bw(reduce, row, rowptr, col, ones if val is None else np.abs(val), np.abs(G), N)
    else:
        _, gB = oracle.spmm_minmax_bw(col, val, B, G, arg, want_value=False)
        _, scale = oracle.spmm_minmax_bw(col, None if val is None else np.abs(val), np.abs(B), np.abs(G), arg, want_value=False)
    got = Bt.grad.float().cpu().numpy()
    assert np.all(np.abs(got - gB) <= 1e-5 * scale + eps * np.abs(gB) + 1e-30), (M, N, K, reduce, half)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_spmm_half_width_trained_values_random_shapes(seed):
    """Trained edge values with bf16 / fp16 dense operands on whatever shape comes (round 4: the half-width passes take
    long columns in chunk waves, the edge-range forward leaves the row-local arg_out in half width too, so every
    reduction stays half width on every matrix whose rows hold at most 65 535 entries): forward and both gradients
    against the oracle on the rounded operands."""
    import paddle_sparse_amd.storage as st_mod
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(5000 + seed)
    M, N, row, col = random_graph(rng)
    K = int(rng.choice([8, 16, 24, 32, 64, 72, 128, 136, 256, 264, 512]))
    reduce = ["sum", "mean", "min", "max"][seed % 4]
    dt = [torch.bfloat16, torch.float16][(seed // 4) % 2]
    eps, tiny = (2.0 ** -8, 0.0) if dt == torch.bfloat16 else (2.0 ** -11, 2.0 ** -25)
    nnz = row.size
    val = rng.standard_normal(nnz).astype(np.float32)
    Bd = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).to(dt)
    Gd = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dt)
    B, G = Bd.float().numpy(), Gd.float().numpy()
    rowptr = oracle.ind2ptr(row, M)
    v = torch.from_numpy(val).cuda().requires_grad_()
    Bt = Bd.cuda().requires_grad_()
    a = SparseTensor(row=idx(row), col=idx(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    old = st_mod.HOT_COLUMNS
    st_mod.HOT_COLUMNS = 64
    try:
        out = a.matmul(Bt, reduce)
        out.backward(Gd.cuda())
    finally:
        st_mod.HOT_COLUMNS = old
    ref, arg = oracle.spmm(reduce, rowptr, col, val, B)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    what = (M, N, K, reduce, dt, nnz)
    assert np.all(np.abs(out.detach().float().cpu().numpy() - ref) <= 1e-5 * S + eps * np.abs(ref) + tiny + 1e-30), what
    if reduce in ("sum", "mean"):
        gB = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, G, N)
        gB_S = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(G), N)
        gV = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
        gV_S = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(B), np.abs(G))
    else:
        gV, gB = oracle.spmm_minmax_bw(col, val, B, G, arg)
        gV_S, gB_S = oracle.spmm_minmax_bw(col, np.abs(val), np.abs(B), np.abs(G), arg)
    assert Bt.grad.dtype == dt and v.grad.dtype == torch.float32
    assert np.all(np.abs(Bt.grad.float().cpu().numpy() - gB) <= 1e-5 * gB_S + eps * np.abs(gB) + tiny + 1e-30), what
    assert np.all(np.abs(v.grad.cpu().numpy() - gV) <= 1e-5 * gV_S + 1e-30), what


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_coalesce_transpose_reduce_random_shapes(seed):
    import paddle_sparse_amd as ps

    rng = np.random.default_rng(2000 + seed)
    M, N = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
    nnz = int(rng.integers(0, 40_000))
    row, col = rng.integers(0, M, nnz), rng.integers(0, N, nnz)
    if seed % 3 == 0 and nnz:  # heavy duplication
        row, col = row % 7, col % 5
    npdtype, tdtype = [(np.float32, torch.float32), (np.int64, torch.int64), (np.float64, torch.float64),
                       (np.int32, torch.int32)][seed % 4]
    shape = [(nnz,), (nnz, 3)][seed % 2]
    val = rng.integers(-9, 10, shape).astype(npdtype)
    op = ["add", "mean", "min", "max"][seed % 4]
    index = np.stack([row, col])
    ref_i, ref_v = so.coalesce(index, val, M, N, op)
    got_i, got_v = ps.coalesce(idx(index), torch.from_numpy(val).cuda(), M, N, op)
    assert np.array_equal(got_i.cpu().numpy(), ref_i), (M, N, nnz, op)
    if npdtype in (np.float32, np.float64) and op == "mean":
        np.testing.assert_allclose(got_v.cpu().numpy(), ref_v, rtol=1e-6)
    else:
        assert np.array_equal(got_v.cpu().numpy(), ref_v), (M, N, nnz, op, npdtype)
    tr_i, tr_v = so.transpose(index, val, M, N)
    gt_i, gt_v = ps.transpose(idx(index), torch.from_numpy(val).cuda(), M, N)
    assert np.array_equal(gt_i.cpu().numpy(), tr_i) and np.array_equal(gt_v.cpu().numpy(), tr_v)
    # SparseTensor: t(), caches and row/column reductions of the coalesced matrix
    t = ps.SparseTensor(row=got_i[0].contiguous(), col=got_i[1].contiguous(), value=got_v, sparse_sizes=(M, N),
                        is_sorted=True)
    st = so.Storage(ref_i[0], ref_i[1], ref_v, (M, N), is_sorted=True)
    tt = t.t()
    ot = so.t(st)
    r, c, v = tt.coo()
    assert np.array_equal(r.cpu().numpy(), ot.row) and np.array_equal(c.cpu().numpy(), ot.col)
    assert np.array_equal(v.cpu().numpy(), ot.value)
    for name in ("rowptr", "rowcount", "colptr", "colcount", "csr2csc", "csc2csr"):
        assert np.array_equal(getattr(t.storage, name)().cpu().numpy(), getattr(st, name)()), name
    if npdtype is not np.int32:
        for dim in (0, 1):
            got = getattr(t, "sum" if op in ("add", "mean") else op)(dim).cpu().numpy()
            assert np.array_equal(got, so.reduction(st, dim, "sum" if op in ("add", "mean") else op)), (dim, op)


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_sample_adj_and_spspmm_random_shapes(seed):
    import paddle_sparse_amd as ps
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(3000 + seed)
    M, N, row, col = random_graph(rng)
    rowptr = oracle.ind2ptr(row, M)
    S = int(rng.integers(0, M + 1))
    subset = rng.integers(0, M, S)
    k, replace = [(-1, False), (3, False), (3, True), (17, False)][seed % 4]
    ref = oracle.sample_adj(rowptr, col, subset, k, replace, seed=seed, num_nodes=max(M, N))
    got = ops.sample_adj(idx(rowptr), idx(col), idx(subset), k, replace, seed=seed, num_cols=N)
    for name, g, r in zip(("rowptr", "col", "n_id", "e_id"), got, ref):
        assert np.array_equal(g.cpu().numpy(), r), (name, M, N, S, k, replace)
    # A (M x N) times a random (N x P) sparse matrix
    P = int(rng.integers(1, 2000))
    keyB = np.unique(rng.integers(0, N * P, int(rng.integers(0, 20_000))))
    iA, iB = np.stack([row, col]), np.stack([keyB // P, keyB % P])
    vA = rng.integers(-3, 4, row.size).astype(np.float32)  # small integers: sums are exact in fp32
    vB = rng.integers(-3, 4, keyB.size).astype(np.float32)
    ref_i, ref_v = oracle.spspmm(iA, vA, iB, vB, M, N, P)
    got_i, got_v = ps.spspmm(idx(iA), torch.from_numpy(vA).cuda(), idx(iB), torch.from_numpy(vB).cuda(), M, N, P)
    assert np.array_equal(got_i.cpu().numpy(), ref_i) and np.array_equal(got_v.cpu().numpy(), ref_v), (M, N, P)


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_add_mul_to_symmetric_random_shapes(seed):
    """A + B, A * B and to_symmetric against the numpy oracle, bit for bit —
    sizes on both sides of the one-workgroup / merge-path switch, operands of
    different shapes, empty operands, overlapping and disjoint patterns."""
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(4000 + seed)
    dtype = [np.float32, np.float64, np.int32, np.int64][seed % 4]

    def make():
        M, N, row, col = random_graph(rng)
        if seed % 3 == 0:  # bigger operands: past the merge-path threshold for to_symmetric too
            extra = np.unique(rng.integers(0, M * N, 60_000))
            key = np.union1d(row * N + col, extra)
            row, col = key // N, key % N
        val = rng.integers(-3, 4, row.size).astype(dtype)
        val[val == 0] = 1
        t = SparseTensor(row=idx(row), col=idx(col), value=torch.from_numpy(val).cuda(), sparse_sizes=(M, N))
        return t, so.Storage(row, col, val, (M, N), is_sorted=True)

    def same(t, o, what):
        row, col, value = t.coo()
        assert t.sparse_sizes() == (o.M, o.N), what
        assert np.array_equal(row.cpu().numpy(), o.row) and np.array_equal(col.cpu().numpy(), o.col), what
        assert np.array_equal(value.cpu().numpy(), o.value), what

    (A, oa), (B, ob) = make(), make()
    same(A + B, so.add(oa, ob), "add")
    same(A + A, so.add(oa, oa), "add self")
    same(A * B, so.mul(oa, ob), "mul")
    same(A * A, so.mul(oa, oa), "mul self")
    for reduce in ("sum", "min", "max"):
        same(A.to_symmetric(reduce), so.to_symmetric(oa, reduce), f"to_symmetric {reduce}")
