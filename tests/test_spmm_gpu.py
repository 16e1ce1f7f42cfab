"""GPU parity: psa_spmm (sum/mean/min/max) through the C-ABI vs the oracle.

Tolerance (north-star: 1e-5 relative for fp32 reductions): per element
|gpu - oracle| <= 1e-5 * S, S = sum_e |w_e * mat[col[e],k]| (the quantity the
rounding error of either summation order is relative to).  arg_out of
min/max is an index and must match bit-exactly whenever the winner is unique.
"""
import numpy as np
import pytest
import torch

import oracle
from util import random_csr, skewed_csr

pytestmark = pytest.mark.gpu

RTOL = 1e-5

# Every test of this file runs on both forward kernel families (psa_spmm_algo):
# one wave per row, and edge ranges — the latter with the COO row ids handed in
# (SparseStorage.row()) and with row = NULL (derived from rowptr by the call).
ALGO = "row_waves"


@pytest.fixture(autouse=True, params=["row_waves", "edge_ranges", "edge_ranges_norow"])
def algo(request):
    global ALGO
    ALGO = request.param
    yield request.param
    ALGO = "row_waves"


def once():
    """For tests that pin kernel variants themselves or do not go through run_gpu."""
    if ALGO != "row_waves":
        pytest.skip("independent of the algo parameter")


def algo_kwargs(rowptr_dev, nnz):
    """algo / row arguments of the current parameter for direct ops.spmm_* calls."""
    from paddle_sparse_amd import ops

    if ALGO == "row_waves":
        return {"algo": "row_waves"}
    return {"algo": "edge_ranges", "row": ops.ptr2ind(rowptr_dev, nnz) if ALGO == "edge_ranges" else None}


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_gpu(reduce, rowptr, col, val, B):
    from paddle_sparse_amd import ops

    fn = getattr(ops, f"spmm_{reduce}")
    row = None
    if ALGO == "edge_ranges":
        row = dev(np.repeat(np.arange(rowptr.size - 1, dtype=np.int64), np.diff(rowptr)))
    res = fn(dev(rowptr), dev(col), dev(val), dev(B), row=row, algo=ALGO.replace("_norow", ""))
    torch.cuda.synchronize()
    if isinstance(res, tuple):
        return res[0].cpu().numpy(), res[1].cpu().numpy()
    return res.cpu().numpy(), None


def check(reduce, rowptr, col, val, B):
    out, arg = run_gpu(reduce, rowptr, col, val, B)
    ref, ref_arg = oracle.spmm(reduce, rowptr, col, val, B)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    err = np.abs(out.astype(np.float64) - ref.astype(np.float64))
    assert np.all(err <= RTOL * S + 1e-30), f"max err ratio {np.max(err / (S + 1e-30))}"
    if reduce in ("min", "max"):
        nnz = col.size
        # values the winners point at must reproduce out exactly
        w = np.ones(nnz, np.float32) if val is None else val
        valid = arg != nnz
        assert np.array_equal(valid, ref_arg != nnz)
        k_idx = np.broadcast_to(np.arange(B.shape[1]), arg.shape)
        picked = w[arg[valid]] * B[col[arg[valid]], k_idx[valid]]
        assert np.array_equal(picked, out[valid])
        assert np.all(out[~valid] == 0)
        assert np.array_equal(arg, ref_arg)
    return out


def test_readme_kat(kats):
    k = kats["spmm"]
    row, col = np.array(k["index"], np.int64)
    rowptr = oracle.ind2ptr(row, k["m"])
    out, _ = run_gpu("sum", rowptr, col, np.array(k["value"], np.float32), np.array(k["matrix"], np.float32))
    assert out.tolist() == k["out"]


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("K", [1, 2, 3, 4, 8, 12, 16, 32, 33, 64, 100, 128, 256, 260, 512])
def test_random_k(reduce, K):
    M, N, nnz = 777, 555, 6000
    row, rowptr, col, val = random_csr(M, N, nnz, seed=K)
    B = np.random.default_rng(K + 1).standard_normal((N, K)).astype(np.float32)
    check(reduce, rowptr, col, val, B)


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [4, 64, 128])
def test_rows_where_nothing_beats_the_init_keep_the_sentinel(reduce, K):
    """a row whose products are all -inf (max) / +inf (min) has no winner.  The oracle
    (upstream: `if (x > acc)`) leaves arg_out = nnz and out = -/+FLT_MAX there; both kernel families
    must too — the edge-range walk once kept the previous row's winner in its registers."""
    from paddle_sparse_amd import ops

    M, N = 40, 30
    rng = np.random.default_rng(K)
    deg = rng.integers(1, 6, M)
    deg[7] = 0
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = rng.integers(0, N - 2, nnz).astype(np.int64)
    dead_rows = [3, 4, 11, M - 1]  # 3 and 4: consecutive, so "stale" would come from a dead row as well
    for r in dead_rows:
        col[rowptr[r]:rowptr[r + 1]] = N - 1 - (r & 1)  # these rows gather only the infinite rows of B
    val = np.abs(rng.standard_normal(nnz)).astype(np.float32) + 0.5  # positive: inf keeps its sign
    B = rng.standard_normal((N, K)).astype(np.float32)
    B[N - 2:] = -np.inf if reduce == "max" else np.inf
    out, arg = run_gpu(reduce, rowptr, col, val, B)
    ref, ref_arg = oracle.spmm(reduce, rowptr, col, val, B)
    assert np.array_equal(arg, ref_arg) and np.array_equal(out, ref)
    assert np.all(arg[dead_rows] == nnz)
    # the row-local byte form agrees between the families as well
    kw = algo_kwargs(dev(rowptr), nnz)
    if K % 4 == 0:
        for width in (1, 2):
            res = ops._spmm(reduce, dev(rowptr), dev(col), dev(val), dev(B), want_arg_bytes=width, **kw)
            base = ops._spmm(reduce, dev(rowptr), dev(col), dev(val), dev(B), want_arg_bytes=width, algo="row_waves")
            assert torch.equal(res[2], base[2]) and torch.equal(res[1], base[1])


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_no_value(reduce):
    row, rowptr, col, _ = random_csr(500, 400, 4000, seed=3, with_value=False)
    B = np.random.default_rng(7).standard_normal((400, 64)).astype(np.float32)
    check(reduce, rowptr, col, None, B)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("K", [4, 64, 128])
def test_skewed_and_empty_rows(reduce, K):
    row, rowptr, col, val = skewed_csr(400, 300, seed=K, long_rows=(0, 5, 399), long_deg=777)
    B = np.random.default_rng(2).standard_normal((300, K)).astype(np.float32)
    check(reduce, rowptr, col, val, B)


def test_all_rows_empty():
    rowptr = np.zeros(101, np.int64)
    col = np.zeros(0, np.int64)
    B = np.ones((10, 8), np.float32)
    for reduce in ("sum", "mean", "min", "max"):
        out, arg = run_gpu(reduce, rowptr, col, np.zeros(0, np.float32), B)
        assert np.all(out == 0)
        if arg is not None:
            assert np.all(arg == 0)  # sentinel = nnz = 0


def test_ties_pick_first_edge():
    """Equal candidates: arg_out is the first winner in edge order."""
    rowptr = np.array([0, 6], np.int64)
    col = np.array([1, 0, 1, 0, 1, 0], np.int64)
    val = np.ones(6, np.float32)
    B = np.array([[5.0] * 128, [5.0] * 128], np.float32)
    for reduce in ("min", "max"):
        out, arg = run_gpu(reduce, rowptr, col, val, B)
        assert np.all(out == 5.0) and np.all(arg == 0)


@pytest.mark.parametrize("variant", [0, 2, 3, 4, 16, 17, 18])  # 17: ordinary stores, 18: non-temporal gathers
def test_variants_k128(variant):
    from paddle_sparse_amd import ops

    once()

    row, rowptr, col, val = random_csr(3000, 2000, 40000, seed=variant)
    B = np.random.default_rng(5).standard_normal((2000, 128)).astype(np.float32)
    prev = ops.spmm_set_variant(variant)
    try:
        for reduce in ("sum", "mean", "max"):
            check(reduce, rowptr, col, val, B)
        if variant >= 16:  # memory-path variants of the production kernel: same arithmetic, same bits
            got = run_gpu("max", rowptr, col, val, B)
            ops.spmm_set_variant(0)
            ref = run_gpu("max", rowptr, col, val, B)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    finally:
        ops.spmm_set_variant(prev)


@pytest.mark.parametrize("variant", [0, 1, 7, 11, 12, 13])
@pytest.mark.parametrize("K", [4, 8, 12, 16, 32, 48, 64, 128, 200, 256])
def test_multirow_variants_narrow_k(variant, K):
    from paddle_sparse_amd import ops

    once()

    row, rowptr, col, val = skewed_csr(1000, 700, seed=K, long_rows=(0, 500, 999), long_deg=300)
    B = np.random.default_rng(K).standard_normal((700, K)).astype(np.float32)
    prev = ops.spmm_set_variant(variant)
    try:
        for reduce in ("sum", "mean", "min", "max"):
            check(reduce, rowptr, col, val, B)
        check("sum", rowptr, col, None, B)
    finally:
        ops.spmm_set_variant(prev)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("K", [4, 32, 64, 100, 128, 256, 300, 384, 512])
def test_long_rows_take_the_chunked_path(reduce, K):
    """Rows above 512 edges are split into 256-edge chunks reduced by separate
    waves and folded in chunk order; results must match the single-wave path
    (variant 10 switches the chunked path off) and the oracle."""
    from paddle_sparse_amd import ops

    once()
    row, rowptr, col, val = skewed_csr(600, 500, seed=K, long_rows=(0, 1, 300, 599), long_deg=5000)
    deg = rowptr[1:] - rowptr[:-1]
    assert (deg > 512).sum() == 4 and (deg == 0).sum() > 0
    B = np.random.default_rng(K + 3).standard_normal((500, K)).astype(np.float32)
    out = check(reduce, rowptr, col, val, B)
    prev = ops.spmm_set_variant(15)  # chunk and row launches back to back (0 fuses them for 64 < K <= 256)
    try:
        separate = check(reduce, rowptr, col, val, B)
    finally:
        ops.spmm_set_variant(prev)
    if K >= 192:
        # production runs these as ceil(K / 128) tiles of 32 lanes (two edges per gather step, folded
        # across the two lane groups); the separate launches use one 64-lane tile: other order
        assert np.all(np.abs(separate - out) <= RTOL * oracle.spmm_abs_sum(rowptr, col, val, B) + 1e-30)
    else:
        assert np.array_equal(separate, out)  # same chunking, same fold order
    prev = ops.spmm_set_variant(10)
    try:
        ref_out, ref_arg = run_gpu(reduce, rowptr, col, val, B)
    finally:
        ops.spmm_set_variant(prev)
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    assert np.all(np.abs(out - ref_out) <= RTOL * S + 1e-30)
    if reduce in ("min", "max"):
        assert np.array_equal(run_gpu(reduce, rowptr, col, val, B)[1], ref_arg)


def test_one_row_holds_everything():
    from paddle_sparse_amd import ops

    nnz, N, K = 300_000, 4000, 64
    rng = np.random.default_rng(0)
    rowptr = np.array([0, 0, nnz, nnz], np.int64)
    col = rng.integers(0, N, nnz, dtype=np.int64)
    val = rng.standard_normal(nnz).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    for reduce in ("sum", "mean", "max"):
        check(reduce, rowptr, col, val, B)


def test_config2_shape_vs_oracle():
    """BASELINE config 2: CSR 100k x 100k, nnz = 1M, F = 64."""
    M = N = 100_000
    row, rowptr, col, val = random_csr(M, N, 1_000_000, seed=1)
    B = np.random.default_rng(1).standard_normal((N, 64)).astype(np.float32)
    check("sum", rowptr, col, val, B)


def test_config4_per_gpu_shape():
    """BASELINE config 4, one rank's share: 2M rows x 16M columns, nnz 20M,
    F = 256 (B is 16.4 GB: exercises 64-bit addressing of the gather).  A
    sample of rows is recomputed in float64 with torch ops."""
    from paddle_sparse_amd import ops

    M, N, nnz, K = 2_000_000, 16_000_000, 20_000_000, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    row = torch.sort(torch.randint(0, M, (nnz,), generator=g, device="cuda"))[0]
    col = torch.randint(0, N, (nnz,), generator=g, device="cuda")
    col[-1] = N - 1  # touch the last row of B
    val = torch.randn(nnz, generator=g, device="cuda")
    rowptr = ops.ind2ptr(row, M)
    B = torch.randn(N, K, generator=g, device="cuda")
    if ALGO == "edge_ranges_norow":
        pytest.skip("same kernels as edge_ranges")
    out = ops.spmm_sum(rowptr, col, val, B, **algo_kwargs(rowptr, nnz))
    sample = torch.cat([torch.arange(0, 500, device="cuda"), torch.arange(M - 500, M, device="cuda")])
    e0 = int(rowptr[500])
    e1 = int(rowptr[M - 500])
    for lo, hi, rows in ((0, e0, sample[:500]), (e1, nnz, sample[500:])):
        contrib = val[lo:hi, None].double() * B[col[lo:hi]].double()
        acc = torch.zeros(500, K, dtype=torch.float64, device="cuda")
        acc.index_add_(0, row[lo:hi] - rows[0], contrib)
        scale = torch.zeros(500, K, dtype=torch.float64, device="cuda").index_add_(0, row[lo:hi] - rows[0], contrib.abs())
        assert bool(((out[rows].double() - acc).abs() <= 1e-5 * scale + 1e-30).all())


def test_linearity_full_size():
    """BASELINE config 3 size (2M x 2M, nnz 20M, F=128): size-independent
    properties instead of the oracle — linearity in B, and A @ ones == row sums."""
    from paddle_sparse_amd import ops

    M = N = 2_000_000
    nnz, K = 20_000_000, 128
    g = torch.Generator(device="cuda").manual_seed(2)
    row = torch.sort(torch.randint(0, M, (nnz,), generator=g, device="cuda"))[0]
    col = torch.randint(0, N, (nnz,), generator=g, device="cuda")
    val = torch.randn(nnz, generator=g, device="cuda")
    rowptr = ops.ind2ptr(row, M)
    B1 = torch.randn(N, K, generator=g, device="cuda")
    B2 = torch.randn(N, K, generator=g, device="cuda")
    if ALGO == "edge_ranges_norow":
        pytest.skip("same kernels as edge_ranges")
    kw = algo_kwargs(rowptr, nnz)
    o1 = ops.spmm_sum(rowptr, col, val, B1, **kw)
    o2 = ops.spmm_sum(rowptr, col, val, B2, **kw)
    o12 = ops.spmm_sum(rowptr, col, val, B1 + 2 * B2, **kw)
    scale = torch.zeros(M, device="cuda").index_add_(0, row, val.abs())
    err = (o12 - (o1 + 2 * o2)).abs().max(dim=1)[0]
    assert bool((err <= 1e-4 * (scale * 6 + 1e-6)).all())
    ones = torch.ones(N, K, device="cuda")
    rs = torch.zeros(M, device="cuda", dtype=torch.float64).index_add_(0, row, val.double())
    o = ops.spmm_sum(rowptr, col, val, ones, **kw)
    assert bool(((o[:, 0].double() - rs).abs() <= 1e-5 * scale.double() + 1e-6).all())
    assert bool((o == o[:, :1]).all())
    # mean / max agree with sum-derived quantities
    deg = (rowptr[1:] - rowptr[:-1]).clamp(min=1).float()
    om = ops.spmm_mean(rowptr, col, val, B1, **kw)
    assert bool(((om - o1 / deg[:, None]).abs() <= 1e-5 * (scale / deg * 6)[:, None] + 1e-6).all())


def test_row_stats_and_algo_choice():
    """psa_csr_row_stats against numpy, and the per-matrix choice built on it."""
    from paddle_sparse_amd import SparseTensor, ops

    once()

    row, rowptr, col, val = skewed_csr(5000, 300, seed=9, long_rows=(0, 77), long_deg=1500, base_deg=1)
    deg = np.diff(rowptr)
    want = (int((deg == 0).sum()), int(((deg >= 1) & (deg <= 2)).sum()), int((deg > 128).sum()), int(deg.max()))
    assert ops.csr_row_stats(dev(rowptr)) == want
    skew = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(5000, 300),
                        is_sorted=True, trust_data=True)
    assert skew.storage._spmm_algo() == "edge_ranges" and skew.storage._longest_row() == want[3]
    row, rowptr, col, val = random_csr(800, 300, 8000, seed=1)
    flat = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(800, 300),
                        is_sorted=True, trust_data=True)
    assert flat.storage._spmm_algo() == "row_waves"
    assert ops.csr_row_stats(dev(np.zeros(1, np.int64))) == (0, 0, 0, 0)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_edge_ranges_boundaries(reduce):
    """Rows cut by range boundaries in every way: one row spanning many ranges, rows
    ending exactly at a boundary, single-edge rows around boundaries, a tail of empty rows."""
    rng = np.random.default_rng(11)
    deg = np.concatenate([[256, 0, 256, 1, 255, 1000, 0, 0, 3, 509, 1, 1, 1, 253], rng.integers(0, 4, 300), [0] * 70])
    M, N, K = deg.size, 97, 128
    rowptr = np.zeros(M + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    col = rng.integers(0, N, nnz, dtype=np.int64)
    val = rng.standard_normal(nnz).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    B[::3] = 1.0  # ties
    check(reduce, rowptr, col, val, B)


def test_edge_ranges_bitwise_reproducible():
    from paddle_sparse_amd import ops

    once()

    row, rowptr, col, val = skewed_csr(3000, 500, seed=4, long_rows=(1, 2000), long_deg=3000, base_deg=2)
    B = np.random.default_rng(3).standard_normal((500, 128)).astype(np.float32)
    d = [dev(x) for x in (rowptr, col, val, B, row)]
    a = ops.spmm_sum(d[0], d[1], d[2], d[3], row=d[4], algo="edge_ranges")
    for _ in range(3):
        assert torch.equal(a, ops.spmm_sum(d[0], d[1], d[2], d[3], row=d[4], algo="edge_ranges"))
    # min/max: identical to the row-wave kernels, values and winners
    for red in ("min", "max"):
        o1, a1 = getattr(ops, f"spmm_{red}")(d[0], d[1], d[2], d[3], algo="row_waves")
        o2, a2 = getattr(ops, f"spmm_{red}")(d[0], d[1], d[2], d[3], row=d[4], algo="edge_ranges")
        assert torch.equal(o1, o2) and torch.equal(a1, a2)


def test_hot_column_copy_gives_the_same_bits():
    """Hub columns: references redirected into a compact copy of their rows of the dense
    operand (psa_spmm_coo hot_rows; SparseStorage._hot_columns).  Same edges, same order,
    same kernel — the result is identical to the plain edge-range forward."""
    from paddle_sparse_amd import SparseTensor, ops
    from paddle_sparse_amd import storage as st_mod

    once()
    rng = np.random.default_rng(17)
    M, N, K = 6000, 5000, 128
    deg = rng.integers(0, 3, M)
    deg[rng.integers(0, M, 30)] = 700  # skewed rows: the edge-range forward is chosen
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = rng.integers(0, N, nnz)
    hubs = rng.integers(0, N, 40)
    pick = rng.random(nnz) < 0.5
    col[pick] = hubs[rng.integers(0, 40, int(pick.sum()))]  # half of all entries point at 40 hub columns
    row = np.repeat(np.arange(M), deg)
    order = np.lexsort((col, row))
    col = col[order].astype(np.int64)
    val = rng.standard_normal(nnz).astype(np.float32)
    B = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    a = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    assert a.storage._spmm_algo() == "edge_ranges"
    old = st_mod.HOT_COLUMNS
    st_mod.HOT_COLUMNS = 256
    try:
        plan = a.storage._hot_columns()
    finally:
        st_mod.HOT_COLUMNS = old
    assert plan is not None
    hot, col_eff = plan
    assert hot.numel() == 256 and set(hubs.tolist()) <= set(hot.tolist())
    back = torch.where(col_eff >= N, hot[(col_eff - N).clamp(min=0)], col_eff)
    assert torch.equal(back, dev(col))
    for reduce in ("sum", "mean", "min", "max"):
        plain = ops._spmm(reduce, a.storage.rowptr(), a.storage.col(), a.storage.value(), B, row=a.storage.row(),
                          algo="edge_ranges")
        hot_rows = ops.gather_rows(B, hot)
        redirected = ops._spmm(reduce, a.storage.rowptr(), col_eff, a.storage.value(), B, row=a.storage.row(),
                               algo="edge_ranges", hot_rows=hot_rows)
        assert torch.equal(plain[0], redirected[0])
        if plain[1] is not None:
            assert torch.equal(plain[1], redirected[1])
        via_api = a.matmul(B, reduce)  # the tensor surface takes the plan by itself
        assert torch.equal(via_api, plain[0])
    with pytest.raises(Exception, match="edge-range"):
        ops._spmm("sum", a.storage.rowptr(), col_eff, a.storage.value(), B, algo="row_waves", hot_rows=hot_rows)
    # a matrix without hub columns builds no plan
    row2, rowptr2, col2, val2 = skewed_csr(3000, 2000, seed=5, long_rows=(1,), long_deg=2000, base_deg=1)
    flat = SparseTensor(rowptr=dev(rowptr2), col=dev(col2), value=dev(val2), sparse_sizes=(3000, 2000), is_sorted=True,
                        trust_data=True)
    assert flat.storage._spmm_algo() == "edge_ranges" and flat.storage._hot_columns() is None
