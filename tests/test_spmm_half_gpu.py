"""GPU parity: psa_spmm_half (fp16 / bf16 dense operands, fp32 accumulation)
against the fp32 oracle run on the SAME 2-byte-rounded inputs.

Tolerance, stated: the kernel's fp32 sum is within 1e-5 * S of the oracle's
(S = sum of the absolute terms, the north-star bar for fp32 reductions) and is
then rounded ONCE to the 2-byte type: |got - ref| <= 1e-5 * S + eps * |ref| with
eps = 2^-8 for bf16 (8 significant bits), 2^-11 for fp16 (plus half the fp16
subnormal spacing, 2^-25, where |ref| < 6.1e-5).  For min / max the
winner's product is exact in fp32, so out == round(ref) and arg_out is identical
wherever the fp32 winner is unique."""
import numpy as np
import pytest
import torch

import oracle
from util import random_csr, skewed_csr

pytestmark = pytest.mark.gpu

EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
TINY = {torch.bfloat16: 0.0, torch.float16: 2.0 ** -25}  # half the subnormal spacing of the 2-byte type


def rounded(a, dtype):
    """float32 numpy array -> (torch 2-byte tensor on the GPU, its exact float32 image)."""
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return t.cuda(), t.float().numpy()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("K", [8, 16, 64, 128, 136, 256, 520])
@pytest.mark.parametrize("variant", [0, 1, 2])  # one row per wave / several rows per wave (K <= 128) / U = 8
def test_half_vs_oracle_on_rounded_inputs(dtype, reduce, K, variant):
    from paddle_sparse_amd import _lib, ops

    if variant != 0 and K > 128:
        pytest.skip("same kernel as variant 0")
    _lib.load().psa_spmm_half_set_variant(variant)
    try:
        _half_case(dtype, reduce, K)
    finally:
        _lib.load().psa_spmm_half_set_variant(0)


def _half_case(dtype, reduce, K):
    from paddle_sparse_amd import ops

    M, N = 700, 500
    row, rowptr, col, val = skewed_csr(M, N, seed=K, long_rows=(0, 350), long_deg=900)
    B = np.random.default_rng(K).standard_normal((N, K)).astype(np.float32)
    Bd, Bf = rounded(B, dtype)
    d = lambda x: torch.from_numpy(x).cuda()  # noqa: E731
    for vmode in ("f32", "half", "none"):
        if vmode == "f32":
            vd, vf = d(val), val
        elif vmode == "half":
            vd, vf = rounded(val, dtype)
        else:
            vd, vf = None, None
        out, arg = ops._spmm(reduce, d(rowptr), d(col), vd, Bd)
        assert out.dtype == dtype
        ref, ref_arg = oracle.spmm(reduce, rowptr, col, vf, Bf)
        S = oracle.spmm_abs_sum(rowptr, col, vf, Bf)
        got = out.float().cpu().numpy()
        assert np.all(np.abs(got - ref) <= 1e-5 * S + EPS[dtype] * np.abs(ref) + TINY[dtype] + 1e-30), (vmode, reduce)
        if reduce in ("min", "max"):
            want = torch.from_numpy(ref).to(dtype).float().numpy()  # one rounding of the exact winner
            assert np.array_equal(got, want)
            assert np.array_equal(arg.cpu().numpy(), ref_arg)


def test_half_falls_back_for_odd_widths_and_keeps_autograd():
    from paddle_sparse_amd import SparseTensor, ops

    row, rowptr, col, val = random_csr(300, 200, 2500, seed=1, sort_cols=True)
    keep = np.concatenate([[True], (row[1:] != row[:-1]) | (col[1:] != col[:-1])])
    row, col, val = row[keep], col[keep], val[keep]
    rowptr = oracle.ind2ptr(row, 300)
    d = lambda x: torch.from_numpy(x).cuda()  # noqa: E731
    B = np.random.default_rng(0).standard_normal((200, 12)).astype(np.float32)  # K % 8 != 0: widened path
    Bd, Bf = rounded(B, torch.bfloat16)
    out, _ = ops._spmm("sum", d(rowptr), d(col), d(val), Bd)
    ref, _ = oracle.spmm("sum", rowptr, col, val, Bf)
    S = oracle.spmm_abs_sum(rowptr, col, val, Bf)
    assert out.dtype == torch.bfloat16
    assert np.all(np.abs(out.float().cpu().numpy() - ref) <= 1e-5 * S + 2.0 ** -8 * np.abs(ref) + 1e-30)
    # autograd: forward in bf16, gradients computed by the fp32 kernels on widened operands
    B = np.random.default_rng(2).standard_normal((200, 32)).astype(np.float32)
    G = np.random.default_rng(3).standard_normal((300, 32)).astype(np.float32)
    Bd, Bf = rounded(B, torch.bfloat16)
    Gd, Gf = rounded(G, torch.bfloat16)
    Bt = Bd.clone().requires_grad_()
    v = d(val).requires_grad_()
    a = SparseTensor(row=d(row), col=d(col), value=v, sparse_sizes=(300, 200), is_sorted=True)
    a.matmul(Bt, "sum").backward(Gd)
    gB = oracle.spmm_mat_bw("sum", row, rowptr, col, val, Gf, 200)
    sB = oracle.spmm_mat_bw("sum", row, rowptr, col, np.abs(val), np.abs(Gf), 200)
    gV = oracle.spmm_value_bw("sum", row, rowptr, col, Bf, Gf)
    sV = oracle.spmm_value_bw("sum", row, rowptr, col, np.abs(Bf), np.abs(Gf))
    assert Bt.grad.dtype == torch.bfloat16 and v.grad.dtype == torch.float32
    assert np.all(np.abs(Bt.grad.float().cpu().numpy() - gB) <= 1e-5 * sB + 2.0 ** -8 * np.abs(gB) + 1e-30)
    assert np.all(np.abs(v.grad.cpu().numpy() - gV) <= 1e-5 * sV + 1e-30)


def test_half_full_size_properties():
    """BASELINE config 3 size in bf16: A @ ones == row sums of the values, linearity in B
    up to the stated rounding."""
    from paddle_sparse_amd import ops

    M = N = 2_000_000
    nnz, K = 20_000_000, 128
    g = torch.Generator(device="cuda").manual_seed(2)
    row = torch.sort(torch.randint(0, M, (nnz,), generator=g, device="cuda"))[0]
    col = torch.randint(0, N, (nnz,), generator=g, device="cuda")
    val = torch.randn(nnz, generator=g, device="cuda")
    rowptr = ops.ind2ptr(row, M)
    ones = torch.ones(N, K, device="cuda", dtype=torch.bfloat16)
    o = ops._spmm("sum", rowptr, col, val, ones)[0]
    rs = torch.zeros(M, device="cuda", dtype=torch.float64).index_add_(0, row, val.double())
    scale = torch.zeros(M, device="cuda", dtype=torch.float64).index_add_(0, row, val.abs().double())
    assert bool(((o[:, 0].double() - rs).abs() <= 1e-5 * scale + 2.0 ** -8 * rs.abs() + 1e-30).all())
    assert bool((o == o[:, :1]).all())
    B = torch.randn(N, K, generator=g, device="cuda").to(torch.bfloat16)
    o1 = ops._spmm("sum", rowptr, col, val, B)[0].float()
    o32 = ops.spmm_sum(rowptr, col, val, B.float())  # the fp32 kernel on the same (widened) operand
    S = ops.spmm_sum(rowptr, col, val.abs(), B.float().abs())
    assert bool(((o1 - o32).abs() <= 2e-5 * S + 2.0 ** -8 * o32.abs() + 1e-30).all())


# ---- the edge-range kernels with 2-byte operands (psa_spmm_half_coo) ------------------------------

@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("K", [8, 16, 32, 64, 128, 256, 264, 520])
@pytest.mark.parametrize("give_row", [True, False])
def test_half_edge_ranges_vs_oracle(dtype, reduce, K, give_row):
    """algo="edge_ranges" with fp16 / bf16 operands: rows of 0, a few, 200 and 900 entries (rows
    crossing many ranges, ranges holding many rows, empty rows between them), fp32 or no values,
    against the oracle on the rounded inputs; min / max exact with identical arg_out; two
    launches give the same bits; the hub-row copy gives the same bits."""
    from paddle_sparse_amd import ops

    M, N = 700, 500
    d = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    for seed, long_rows, long_deg in ((K, (0, 350, 699), 900), (K + 1, (3,), 200)):
        row, rowptr, col, val = skewed_csr(M, N, seed=seed, long_rows=long_rows, long_deg=long_deg)
        B = np.random.default_rng(K).standard_normal((N, K)).astype(np.float32)
        Bd, Bf = rounded(B, dtype)
        for v in (val, None):
            out, arg = ops._spmm(reduce, d(rowptr), d(col), d(v), Bd, row=d(row) if give_row else None, algo="edge_ranges")
            assert out.dtype == dtype
            ref, ref_arg = oracle.spmm(reduce, rowptr, col, v, Bf)
            S = oracle.spmm_abs_sum(rowptr, col, v, Bf)
            got = out.float().cpu().numpy()
            assert np.all(np.abs(got - ref) <= 1e-5 * S + EPS[dtype] * np.abs(ref) + TINY[dtype] + 1e-30), (reduce, v is None)
            if reduce in ("min", "max"):
                assert np.array_equal(got, torch.from_numpy(ref).to(dtype).float().numpy())
                assert np.array_equal(arg.cpu().numpy(), ref_arg)
                plain = ops._spmm(reduce, d(rowptr), d(col), d(v), Bd)  # one wave per row: same results exactly
                assert torch.equal(plain[0], out) and torch.equal(plain[1], arg)
            again = ops._spmm(reduce, d(rowptr), d(col), d(v), Bd, row=d(row) if give_row else None, algo="edge_ranges")
            assert torch.equal(again[0], out)
            # hub-row copy: references to the hot columns go to a compact copy — same edges, same order
            hot = torch.tensor([7, 0, 499, 123, 77], device="cuda")
            slot = torch.full((N,), -1, dtype=torch.int64, device="cuda")
            slot[hot] = torch.arange(hot.numel(), device="cuda")
            c = d(col)
            col_eff = torch.where(slot[c] >= 0, N + slot[c], c)
            copy = ops._spmm(reduce, d(rowptr), col_eff, d(v), Bd, row=d(row), algo="edge_ranges", hot_rows=Bd[hot].contiguous())
            assert torch.equal(copy[0], out)
            if reduce in ("min", "max"):
                assert torch.equal(copy[1], arg)


def test_half_surface_takes_edge_ranges_on_a_power_law_matrix():
    """SparseTensor.matmul with a bf16 operand on a hub-dominated matrix: edge ranges + hub-row
    copy chosen per matrix (as for fp32), result within the stated bound of the oracle."""
    import paddle_sparse_amd.storage as st_mod
    from paddle_sparse_amd import SparseTensor, ops

    rng = np.random.default_rng(46)
    M, N, K = 6000, 5000, 64
    deg = rng.integers(0, 3, M)
    deg[rng.integers(0, M, 20)] = 800
    row = np.repeat(np.arange(M), deg)
    col = rng.integers(0, N, row.size)
    hubs = rng.integers(0, N, 40)
    pick = rng.random(row.size) < 0.6
    col[pick] = hubs[rng.integers(0, 40, int(pick.sum()))]
    key = np.unique(row * N + col)
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)
    Bd, Bf = rounded(rng.standard_normal((N, K)).astype(np.float32), torch.bfloat16)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    a = SparseTensor(row=d(row), col=d(col), value=d(val), sparse_sizes=(M, N), is_sorted=True)
    seen = []
    real = ops._spmm_half

    def spy(*args, **kw):
        seen.append((kw.get("algo"), kw.get("hot_rows") is not None))
        return real(*args, **kw)

    old = st_mod.HOT_COLUMNS
    st_mod.HOT_COLUMNS, ops._spmm_half = 64, spy
    try:
        with torch.no_grad():
            out = a.matmul(Bd, "sum")
    finally:
        st_mod.HOT_COLUMNS, ops._spmm_half = old, real
    assert seen == [("edge_ranges", True)] and out.dtype == torch.bfloat16
    ref, _ = oracle.spmm("sum", rowptr, col, val, Bf)
    S = oracle.spmm_abs_sum(rowptr, col, val, Bf)
    assert np.all(np.abs(out.float().cpu().numpy() - ref) <= 1e-5 * S + 2.0 ** -8 * np.abs(ref) + 1e-30)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("graph", ["uniform", "hubs"])
@pytest.mark.parametrize("with_value", [True, False])
def test_half_backward_of_a_fixed_adjacency_stays_half_width(reduce, graph, with_value):
    """Gradient wrt a bf16 dense operand with fixed edge weights: the half-width forward over
    the CSC view (fp32 weights and sums, one rounding), no fp32 copies of the operands; against
    the oracle on the rounded grad_out with the stated bound."""
    import paddle_sparse_amd.storage as st_mod
    from paddle_sparse_amd import SparseTensor, ops

    rng = np.random.default_rng(48)
    K = 64
    if graph == "uniform":
        M, N = 3000, 2500
        row, rowptr, col, val = random_csr(M, N, 30_000, seed=7, sort_cols=True)
    else:
        M, N = 6000, 5000
        deg = rng.integers(0, 3, M)
        deg[rng.integers(0, M, 20)] = 800
        row = np.repeat(np.arange(M), deg)
        col = rng.integers(0, N, row.size)
        hubs = rng.integers(0, N, 40)
        pick = rng.random(row.size) < 0.6
        col[pick] = hubs[rng.integers(0, 40, int(pick.sum()))]
        key = np.unique(row * N + col)
        row, col = key // N, key % N
        val = rng.standard_normal(key.size).astype(np.float32)
        rowptr = oracle.ind2ptr(row, M)
    keep = np.concatenate([[True], (row[1:] != row[:-1]) | (col[1:] != col[:-1])])
    row, col, val = row[keep], col[keep], val[keep]
    rowptr = oracle.ind2ptr(row, M)
    if not with_value:
        val = None
    d = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    Bd, _ = rounded(rng.standard_normal((N, K)).astype(np.float32), torch.bfloat16)
    Gd, Gf = rounded(rng.standard_normal((M, K)).astype(np.float32), torch.bfloat16)
    a = SparseTensor(row=d(row), col=d(col), value=d(val), sparse_sizes=(M, N), is_sorted=True)
    Bt = Bd.clone().requires_grad_()
    widened = []
    real = ops._spmm

    def spy(reduce_, rowptr_, col_, value_, mat_, **kw):
        widened.append(mat_.dtype)
        return real(reduce_, rowptr_, col_, value_, mat_, **kw)

    old = st_mod.HOT_COLUMNS
    st_mod.HOT_COLUMNS, ops._spmm = 64, spy
    try:
        a.matmul(Bt, reduce).backward(Gd)
    finally:
        st_mod.HOT_COLUMNS, ops._spmm = old, real
    assert widened == [torch.bfloat16, torch.bfloat16]  # forward and backward, both half-width
    assert Bt.grad.dtype == torch.bfloat16
    ones = np.ones(col.size, np.float32)
    want = oracle.spmm_mat_bw(reduce, row, rowptr, col, ones if val is None else val, Gf, N)
    scale = oracle.spmm_mat_bw(reduce, row, rowptr, col, ones if val is None else np.abs(val), np.abs(Gf), N)
    assert np.all(np.abs(Bt.grad.float().cpu().numpy() - want) <= 1e-5 * scale + 2.0 ** -8 * np.abs(want) + 1e-30)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("K", [8, 64, 128, 200, 512])
@pytest.mark.parametrize("value_dtype", ["f32", "half"])
def test_half_backward_with_trained_values_stays_half_width(dtype, reduce, K, value_dtype):
    """VERDICT r02 #9a: trained edge values + a half-width dense operand.  Both gradients come from ONE
    pass over the CSC view that gathers 2-byte rows of grad_out (psa_spmm_half_sum_bw_csc): no fp32
    copy of mat or grad_out is made (the fp32 kernels are not called at all), grad_mat is the fp32 sum
    rounded once, grad_value is fp32 arithmetic on the rounded operands.  Against the oracle on the
    same rounded inputs: |gm - ref| <= 1e-5 * S + eps * |ref|, |gv - ref| <= 1e-5 * S (cast to the
    value's dtype: + eps * |ref|)."""
    from paddle_sparse_amd import SparseTensor, ops

    rng = np.random.default_rng(K)
    M, N = 3000, 2500
    row, rowptr, col, val = random_csr(M, N, 40_000, seed=K, sort_cols=True)
    keep = np.concatenate([[True], (row[1:] != row[:-1]) | (col[1:] != col[:-1])])
    row, col, val = row[keep], col[keep], val[keep]
    rowptr = oracle.ind2ptr(row, M)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    Bd, Bf = rounded(rng.standard_normal((N, K)).astype(np.float32), dtype)
    Gd, Gf = rounded(rng.standard_normal((M, K)).astype(np.float32), dtype)
    if value_dtype == "half":
        vd, vf = rounded(val, dtype)
    else:
        vd, vf = d(val), val
    v = vd.clone().requires_grad_()
    Bt = Bd.clone().requires_grad_()
    a = SparseTensor(row=d(row), col=d(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    called = []
    real_fp32 = (ops.spmm_sum_bw_csc, ops.spmm_value_bw)
    ops.spmm_sum_bw_csc = lambda *x, **k: called.append("fp32 csc pass") or real_fp32[0](*x, **k)
    ops.spmm_value_bw = lambda *x, **k: called.append("fp32 value pass") or real_fp32[1](*x, **k)
    try:
        a.matmul(Bt, reduce).backward(Gd)
    finally:
        ops.spmm_sum_bw_csc, ops.spmm_value_bw = real_fp32
    assert called == [], called  # no fp32 pass, hence no widened operands
    assert Bt.grad.dtype == dtype and v.grad.dtype == vd.dtype
    eps, tiny = EPS[dtype], TINY[dtype]
    gm_ref = oracle.spmm_mat_bw(reduce, row, rowptr, col, vf, Gf, N)
    gm_S = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(vf), np.abs(Gf), N)
    assert np.all(np.abs(Bt.grad.float().cpu().numpy() - gm_ref) <= 1e-5 * gm_S + eps * np.abs(gm_ref) + tiny + 1e-30)
    gv_ref = oracle.spmm_value_bw(reduce, row, rowptr, col, Bf, Gf)
    gv_S = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(Bf), np.abs(Gf))
    cast = eps * np.abs(gv_ref) + tiny if value_dtype == "half" else 0.0
    assert np.all(np.abs(v.grad.float().cpu().numpy() - gv_ref) <= 1e-5 * gv_S + cast + 1e-30)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [8, 64, 128, 512])
@pytest.mark.parametrize("mode", ["trained f32 values", "trained half values", "fixed adjacency", "no values"])
@pytest.mark.parametrize("long_row", [0, 300])  # 0: one-byte row-local arg_out; 300: a row above 128 entries -> two bytes
def test_half_minmax_backward_stays_half_width(dtype, reduce, K, mode, long_row):
    """VERDICT r02 #9a, min / max: with a half-width dense operand the forward leaves the row-local arg_out only
    (no int64 arg_out) and the backward is ONE masked half-width pass over the CSC view — no fp32 copies of mat /
    grad_out, no fp32 kernel.  Winners are exact (a product of the rounded operands is the same fp32 number on
    both sides), so grad_mat / grad_value are the oracle's on the rounded inputs up to the summation order and
    the one rounding on store."""
    from paddle_sparse_amd import SparseTensor, ops

    rng = np.random.default_rng(K + long_row)
    M, N = 2000, 1500
    row, rowptr, col, val = random_csr(M, N, 24_000, seed=K + 1, sort_cols=True)
    if long_row:
        extra_c = rng.choice(N, long_row, replace=False)
        row = np.concatenate([row, np.full(long_row, 7)])
        col = np.concatenate([col, extra_c])
        val = np.concatenate([val, rng.standard_normal(long_row).astype(np.float32)])
    key = np.unique(row * N + col, return_index=True)[1]
    row, col, val = row[key], col[key], val[key]
    order = np.lexsort((col, row))
    row, col, val = row[order], col[order], val[order]
    rowptr = oracle.ind2ptr(row, M)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    Bd, Bf = rounded(rng.standard_normal((N, K)).astype(np.float32), dtype)
    Gd, Gf = rounded(rng.standard_normal((M, K)).astype(np.float32), dtype)
    if mode == "trained half values":
        vd, vf = rounded(val, dtype)
    elif mode == "no values":
        vd, vf = None, np.ones(col.size, np.float32)
    else:
        vd, vf = d(val), val
    trained = mode.startswith("trained")
    v = None if vd is None else (vd.clone().requires_grad_() if trained else vd)
    Bt = Bd.clone().requires_grad_()
    a = SparseTensor(row=d(row), col=d(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    called = []
    real = (ops.spmm_minmax_bw_csc, ops.spmm_minmax_bw, ops.spmm_minmax_bw_eb)
    ops.spmm_minmax_bw_csc = lambda *x, **k: called.append("fp32 csc") or real[0](*x, **k)
    ops.spmm_minmax_bw = lambda *x, **k: called.append("fp32 atomics") or real[1](*x, **k)
    ops.spmm_minmax_bw_eb = lambda *x, **k: called.append("fp32 eb") or real[2](*x, **k)
    try:
        out = a.matmul(Bt, reduce)
        out.backward(Gd)
    finally:
        ops.spmm_minmax_bw_csc, ops.spmm_minmax_bw, ops.spmm_minmax_bw_eb = real
    assert called == [], called
    ref_out, ref_arg = oracle.spmm(reduce, rowptr, col, vf, Bf)
    eps, tiny = EPS[dtype], TINY[dtype]
    assert np.all(np.abs(out.detach().float().cpu().numpy() - ref_out) <= eps * np.abs(ref_out) + tiny)
    gv_ref, gm_ref = oracle.spmm_minmax_bw(col, vf, Bf, Gf, ref_arg)
    gv_S, gm_S = oracle.spmm_minmax_bw(col, np.abs(vf), np.abs(Bf), np.abs(Gf), ref_arg)
    assert Bt.grad.dtype == dtype
    assert np.all(np.abs(Bt.grad.float().cpu().numpy() - gm_ref) <= 1e-5 * gm_S + eps * np.abs(gm_ref) + tiny + 1e-30)
    if trained:
        assert v.grad.dtype == vd.dtype
        cast = eps * np.abs(gv_ref) + tiny if mode == "trained half values" else 0.0
        assert np.all(np.abs(v.grad.float().cpu().numpy() - gv_ref) <= 1e-5 * gv_S + cast + 1e-30)


def _hub_column_matrix(M, N, seed):
    """Random entries plus hub COLUMNS of 2 600, 700 and 129 entries and a 300-entry row: the CSC view has long columns
    (chunks of 128 entries with a wave each, psa_spmm_half_*_bw_csc's workspace path) and a last chunk of one entry."""
    rng = np.random.default_rng(seed)
    row, _, col, val = random_csr(M, N, 20_000, seed=seed, sort_cols=True)
    extra_r, extra_c = [np.full(300, 11)], [rng.choice(N, 300, replace=False)]
    for c, n in ((5, 2600), (N - 1, 700), (N // 2, 129)):
        extra_r.append(rng.choice(M, n, replace=False))
        extra_c.append(np.full(n, c))
    row = np.concatenate([row] + extra_r)
    col = np.concatenate([col] + extra_c)
    key = np.unique(row * N + col, return_index=True)[1]
    row, col = row[key], col[key]
    order = np.lexsort((col, row))
    row, col = row[order], col[order]
    val = rng.standard_normal(row.size).astype(np.float32)
    return row, oracle.ind2ptr(row, M), col, val


def _power_law_matrix(M, N, seed):
    """Rows and columns drawn with a cubic skew: most rows hold at most two entries (the edge-range forward is chosen),
    a few hold hundreds, and the transpose has long columns."""
    rng = np.random.default_rng(seed)
    n = 30_000
    row = np.minimum((rng.random(n) ** 3 * M).astype(np.int64), M - 1)
    col = np.minimum((rng.random(n) ** 3 * N).astype(np.int64), N - 1)
    key = np.unique(row * N + col)
    row, col = key // N, key % N
    val = rng.standard_normal(row.size).astype(np.float32)
    return row, oracle.ind2ptr(row, M), col, val


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("K", [64, 128])
def test_half_training_step_on_a_power_law_matrix_stays_half_width(dtype, reduce, K):
    """VERDICT r03 #4, the other half: a power-law matrix takes the edge-range FORWARD, which now leaves the row-local
    arg_out behind in half width too (psa_spmm_half_coo(arg_bytes)), so min / max — like sum / mean — train without an
    int64 arg_out and without fp32 copies of the operands: no fp32 pass is called, gradients against the oracle."""
    from paddle_sparse_amd import SparseTensor, ops

    M, N = 20_000, 8_000
    row, rowptr, col, val = _power_law_matrix(M, N, seed=K)
    rng = np.random.default_rng(K + 1)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    Bd, Bf = rounded(rng.standard_normal((N, K)).astype(np.float32), dtype)
    Gd, Gf = rounded(rng.standard_normal((M, K)).astype(np.float32), dtype)
    v = d(val).requires_grad_()
    Bt = Bd.clone().requires_grad_()
    a = SparseTensor(row=d(row), col=d(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    assert a.storage._spmm_algo() == "edge_ranges" and 128 < a.storage._longest_row() <= 65_535
    assert a.storage._csc_view()._longest_row() > 128
    called, wanted = [], []
    names = ("spmm_sum_bw_csc", "spmm_value_bw", "spmm_minmax_bw_csc", "spmm_minmax_bw", "spmm_minmax_bw_eb")
    real = {n: getattr(ops, n) for n in names}
    real_half = ops._spmm_half
    for n in names:
        setattr(ops, n, (lambda n_: lambda *x, **k: called.append(n_) or real[n_](*x, **k))(n))
    ops._spmm_half = lambda *x, **k: wanted.append((k.get("algo"), k.get("want_arg", x[5] if len(x) > 5 else True), k.get("want_arg_bytes", False))) or real_half(*x, **k)
    try:
        out = a.matmul(Bt, reduce)
        out.backward(Gd)
    finally:
        for n in names:
            setattr(ops, n, real[n])
        ops._spmm_half = real_half
    assert called == [], called
    if reduce in ("max", "min"):  # the forward: edge ranges, two-byte row-local winners, no int64 arg_out
        assert wanted[0] == ("edge_ranges", False, 2), wanted
    eps, tiny = EPS[dtype], TINY[dtype]
    ref_out, ref_arg = oracle.spmm(reduce, rowptr, col, val, Bf)
    S = oracle.spmm_abs_sum(rowptr, col, val, Bf) if reduce in ("sum", "mean") else 0.0
    assert np.all(np.abs(out.detach().float().cpu().numpy() - ref_out) <= eps * np.abs(ref_out) + 1e-5 * S + tiny)
    if reduce in ("sum", "mean"):
        gm_ref = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, Gf, N)
        gm_S = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(Gf), N)
        gv_ref = oracle.spmm_value_bw(reduce, row, rowptr, col, Bf, Gf)
        gv_S = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(Bf), np.abs(Gf))
    else:
        gv_ref, gm_ref = oracle.spmm_minmax_bw(col, val, Bf, Gf, ref_arg)
        gv_S, gm_S = oracle.spmm_minmax_bw(col, np.abs(val), np.abs(Bf), np.abs(Gf), ref_arg)
    assert np.all(np.abs(Bt.grad.float().cpu().numpy() - gm_ref) <= 1e-5 * gm_S + eps * np.abs(gm_ref) + tiny + 1e-30)
    assert np.all(np.abs(v.grad.float().cpu().numpy() - gv_ref) <= 1e-5 * gv_S + 1e-30)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
@pytest.mark.parametrize("K", [8, 24, 64, 128, 512])
def test_half_backward_with_hub_columns_stays_half_width(dtype, reduce, K):
    """a transpose with long columns (the hub rows / columns of a power-law graph) no longer sends the
    half-width training step to the fp32 kernels on widened operands: columns above 128 entries of the CSC view are
    cut into chunks with a wave each, the chunks' fp32 partials are added in chunk order and rounded once.  Both
    gradients against the oracle on the rounded inputs, and no fp32 pass is called."""
    from paddle_sparse_amd import SparseTensor, ops

    M, N = 3000, 2500
    row, rowptr, col, val = _hub_column_matrix(M, N, seed=K)
    rng = np.random.default_rng(K + 1)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    Bd, Bf = rounded(rng.standard_normal((N, K)).astype(np.float32), dtype)
    Gd, Gf = rounded(rng.standard_normal((M, K)).astype(np.float32), dtype)
    v = d(val).requires_grad_()
    Bt = Bd.clone().requires_grad_()
    a = SparseTensor(row=d(row), col=d(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    assert a.storage._csc_view()._longest_row() >= 2600
    called = []
    names = ("spmm_sum_bw_csc", "spmm_value_bw", "spmm_minmax_bw_csc", "spmm_minmax_bw", "spmm_minmax_bw_eb")
    real = {n: getattr(ops, n) for n in names}
    for n in names:
        setattr(ops, n, (lambda n_: lambda *x, **k: called.append(n_) or real[n_](*x, **k))(n))
    try:
        out = a.matmul(Bt, reduce)
        out.backward(Gd)
    finally:
        for n in names:
            setattr(ops, n, real[n])
    assert called == [], called
    eps, tiny = EPS[dtype], TINY[dtype]
    if reduce in ("sum", "mean"):
        gm_ref = oracle.spmm_mat_bw(reduce, row, rowptr, col, val, Gf, N)
        gm_S = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(Gf), N)
        gv_ref = oracle.spmm_value_bw(reduce, row, rowptr, col, Bf, Gf)
        gv_S = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(Bf), np.abs(Gf))
    else:
        _, ref_arg = oracle.spmm(reduce, rowptr, col, val, Bf)
        gv_ref, gm_ref = oracle.spmm_minmax_bw(col, val, Bf, Gf, ref_arg)
        gv_S, gm_S = oracle.spmm_minmax_bw(col, np.abs(val), np.abs(Bf), np.abs(Gf), ref_arg)
    assert np.all(np.abs(Bt.grad.float().cpu().numpy() - gm_ref) <= 1e-5 * gm_S + eps * np.abs(gm_ref) + tiny + 1e-30)
    assert np.all(np.abs(v.grad.float().cpu().numpy() - gv_ref) <= 1e-5 * gv_S + 1e-30)


@pytest.mark.parametrize("K", [128])
def test_half_long_column_chunks_match_the_one_wave_walk(K):
    """The same pass with and without the long-column workspace: grad_value is the same bits (every entry's dot is
    computed the same way wherever it runs), grad_mat the same up to the order in which a long column's terms are added."""
    from paddle_sparse_amd import SparseStorage, ops

    M, N = 3000, 2500
    row, rowptr, col, val = _hub_column_matrix(M, N, seed=3)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    st = SparseStorage(row=d(row), rowptr=d(rowptr), col=d(col), value=d(val), sparse_sizes=(M, N), is_sorted=True)
    g = torch.Generator(device="cuda").manual_seed(5)
    B = torch.randn(N, K, generator=g, device="cuda").to(torch.bfloat16)
    G = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = ops.gather_rows(st.value(), st.csr2csc())
    gv1, gm1 = ops.spmm_half_sum_bw_csc(st.colptr(), st._row_in_csc_order(), w, B, G, True, long_columns=True)
    gv0, gm0 = ops.spmm_half_sum_bw_csc(st.colptr(), st._row_in_csc_order(), w, B, G, True, long_columns=False)
    assert torch.equal(gv1, gv0)
    short = (st.colcount() <= 128)
    assert torch.equal(gm1[short], gm0[short])
    assert torch.allclose(gm1.float(), gm0.float(), rtol=2.0 ** -7, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K", [64, 128])
def test_half_64_bit_addressing_gives_the_bits_of_the_32_bit_form(dtype, K):
    """operands below 4 GiB take 32-bit byte offsets (N bounds col and is mat's height — the header says
    so); every committed test shape is that small, so the 64-bit instantiations (what a 4 GiB operand runs: config
    4 in bf16) were never reached.  psa_spmm_half_set_variant(4) forces them: forward (sum, and max with the row-local
    arg_out) and both passes over the CSC view (sum and masked, with grad_value, long columns in chunks) on the same
    data must give the same bits."""
    from paddle_sparse_amd import SparseStorage, _lib, ops

    M, N = 3000, 2500
    row, rowptr, col, val = _hub_column_matrix(M, N, seed=K + 7)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    st = SparseStorage(row=d(row), rowptr=d(rowptr), col=d(col), value=d(val), sparse_sizes=(M, N), is_sorted=True)
    g = torch.Generator(device="cuda").manual_seed(K)
    B = torch.randn(N, K, generator=g, device="cuda").to(dtype)
    G = torch.randn(M, K, generator=g, device="cuda").to(dtype)
    w = ops.gather_rows(st.value(), st.csr2csc())

    def run():
        out_sum = ops._spmm("sum", st.rowptr(), st.col(), st.value(), B)[0]
        out_max, _, words = ops._spmm("max", st.rowptr(), st.col(), st.value(), B, want_arg=False, want_arg_bytes=2)
        gv, gm = ops.spmm_half_sum_bw_csc(st.colptr(), st._row_in_csc_order(), w, B, G, True)
        mv, mm = ops.spmm_half_minmax_bw_csc(st.colptr(), st._row_in_csc_order(), st._csc_edge_tags(2), w, B, G, words)
        return out_sum, out_max, words, gv, gm, mv, mm

    small = run()
    _lib.load().psa_spmm_half_set_variant(4)
    try:
        wide = run()
    finally:
        _lib.load().psa_spmm_half_set_variant(0)
    for a, b in zip(small, wide):
        assert torch.equal(a, b)
