"""GPU: BASELINE config 3 at full size (2 M x 2 M, 20 M entries, F = 128) for
the SpMM results the oracle is too slow for — min / max forward and all three
backwards — through size-independent properties:

  * min / max: out[r, k] == value[arg] * B[col[arg], k] gathered on the device,
    arg inside row r, no candidate of the row beats the winner (sampled rows in
    float64), empty rows give 0 / the sentinel;
  * backward linearity in grad_out, and grad_mat / grad_value against float64
    index_add_ recomputation on row and column samples.
Both forward kernel families where they apply."""
import pytest
import torch

pytestmark = pytest.mark.gpu

M = N = 2_000_000
NNZ, K = 20_000_000, 128


@pytest.fixture(scope="module")
def c3():
    from paddle_sparse_amd import SparseTensor, ops

    g = torch.Generator(device="cuda").manual_seed(2)
    row = torch.sort(torch.randint(0, M, (NNZ,), generator=g, device="cuda"))[0]
    col = torch.randint(0, N, (NNZ,), generator=g, device="cuda")
    val = torch.randn(NNZ, generator=g, device="cuda")
    rowptr = ops.ind2ptr(row, M)
    B = torch.randn(N, K, generator=g, device="cuda")
    G = torch.randn(M, K, generator=g, device="cuda")
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    return dict(row=row, col=col, val=val, rowptr=rowptr, B=B, G=G, a=a, g=g)


@pytest.mark.parametrize("reduce", ["max", "min"])
@pytest.mark.parametrize("algo", ["row_waves", "edge_ranges"])
def test_minmax_forward_properties(c3, reduce, algo):
    from paddle_sparse_amd import ops

    row, col, val, rowptr, B = c3["row"], c3["col"], c3["val"], c3["rowptr"], c3["B"]
    out, arg = getattr(ops, f"spmm_{reduce}")(rowptr, col, val, B, row=row if algo == "edge_ranges" else None, algo=algo)
    deg = rowptr[1:] - rowptr[:-1]
    empty = deg == 0
    assert bool((out[empty] == 0).all()) and bool((arg[empty] == NNZ).all())
    ne = ~empty
    a = arg[ne]
    lo, hi = rowptr[:-1][ne], rowptr[1:][ne]
    assert bool(((a >= lo[:, None]) & (a < hi[:, None])).all())  # the winner is an edge of its own row
    # the winner's product reproduces out exactly (same fp32 multiply)
    rows = torch.nonzero(ne).flatten()
    for part in torch.chunk(torch.arange(rows.numel(), device="cuda"), 8):
        r = rows[part]
        ar = arg[r]
        picked = val[ar] * B[col[ar], torch.arange(K, device="cuda").expand_as(ar)]
        assert torch.equal(picked, out[r])
    # no candidate of a sampled row beats (or, earlier in edge order, ties) the winner
    sample = torch.cat([torch.arange(0, 2000, device="cuda"), torch.arange(M - 2000, M, device="cuda")])
    for r in sample[deg[sample] > 0][::7].tolist():
        s, e = int(rowptr[r]), int(rowptr[r + 1])
        cand = val[s:e, None] * B[col[s:e]]
        best = cand.max(0) if reduce == "max" else cand.min(0)
        assert torch.equal(best.values, out[r])
        first = (cand == best.values[None]).float().argmax(0) + s  # first edge reaching the extreme
        assert torch.equal(first, arg[r])


def _f64_grad_mat_rows(c3, weights, G, cols):
    """grad_mat[c, :] = sum over edges with col == c of weights[e] * G[row[e], :] in float64, for c in cols."""
    row, col = c3["row"], c3["col"]
    pick = torch.isin(col, cols)
    e = torch.nonzero(pick).flatten()
    slot = torch.searchsorted(cols, col[e])
    acc = torch.zeros(cols.numel(), K, dtype=torch.float64, device="cuda")
    acc.index_add_(0, slot, weights[e, None].double() * G[row[e]].double())
    scale = torch.zeros(cols.numel(), K, dtype=torch.float64, device="cuda")
    scale.index_add_(0, slot, (weights[e, None].double() * G[row[e]].double()).abs())
    return acc, scale


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_sum_mean_backward_properties(c3, reduce):
    a, val, B, G, g = c3["a"], c3["val"], c3["B"], c3["G"], c3["g"]
    row, col, rowptr = c3["row"], c3["col"], c3["rowptr"]
    deg = (rowptr[1:] - rowptr[:-1]).clamp(min=1).float()

    def grads(grad_out):
        v = val.clone().requires_grad_()
        Bt = B.clone().requires_grad_()
        a.set_value(v, layout="coo").matmul(Bt, reduce).backward(grad_out)
        return v.grad, Bt.grad

    gv1, gm1 = grads(G)
    G2 = torch.randn(M, K, generator=g, device="cuda")
    gv2, gm2 = grads(G2)
    gv12, gm12 = grads(G + 2 * G2)
    w = val / deg[row] if reduce == "mean" else val
    # linearity in grad_out (scale: sum of absolute terms, from a second pass on absolute operands)
    cols = torch.unique(torch.randint(0, N, (3000,), generator=g, device="cuda"))
    ref, scale = _f64_grad_mat_rows(c3, w, G, cols)
    assert bool(((gm1[cols].double() - ref).abs() <= 1e-5 * scale + 1e-30).all())
    ref2, scale2 = _f64_grad_mat_rows(c3, w, G2, cols)
    assert bool(((gm12[cols].double() - (ref + 2 * ref2)).abs() <= 1e-5 * (scale + 2 * scale2) + 1e-30).all())
    assert bool(((gm12 - (gm1 + 2 * gm2)).abs().amax(1) <= 1e-4 * (gm1.abs() + 2 * gm2.abs()).amax(1) + 1e-4).all())
    # grad_value on an edge sample
    e = torch.randint(0, NNZ, (200_000,), generator=g, device="cuda")
    terms = B[col[e]].double() * G[row[e]].double()
    ref_v = terms.sum(1) / (deg[row[e]].double() if reduce == "mean" else 1.0)
    sc_v = terms.abs().sum(1) / (deg[row[e]].double() if reduce == "mean" else 1.0)
    assert bool(((gv1[e].double() - ref_v).abs() <= 1e-5 * sc_v + 1e-30).all())
    assert bool(((gv12 - (gv1 + 2 * gv2)).abs() <= 1e-4 * (gv1.abs() + 2 * gv2.abs()) + 1e-4).all())
    # values only (trained edge values over fixed input features: a first GNN layer): autograd takes spmm_value_bw, the
    # CSR-side pass that reads grad_out[row] once per row and gathers mat[col] — same numbers as the one-pass kernel
    # up to the order of the K-term dot, and the float64 sample again
    v = val.clone().requires_grad_()
    a.set_value(v, layout="coo").matmul(B, reduce).backward(G)
    assert bool(((v.grad[e].double() - ref_v).abs() <= 1e-5 * sc_v + 1e-30).all())
    assert bool(((v.grad - gv1).abs() <= 2e-5 * (v.grad.abs() + gv1.abs()) + 1e-4).all())


@pytest.mark.parametrize("reduce", ["max", "min"])
def test_minmax_backward_properties(c3, reduce):
    from paddle_sparse_amd import ops

    a, val, B, G, g = c3["a"], c3["val"], c3["B"], c3["G"], c3["g"]
    row, col, rowptr = c3["row"], c3["col"], c3["rowptr"]
    _, arg = getattr(ops, f"spmm_{reduce}")(rowptr, col, val, B)

    def grads(grad_out):
        v = val.clone().requires_grad_()
        Bt = B.clone().requires_grad_()
        a.set_value(v, layout="coo").matmul(Bt, reduce).backward(grad_out)
        return v.grad, Bt.grad

    gv1, gm1 = grads(G)
    # every (row, k) routes grad_out[row, k] to ONE edge: totals are conserved
    valid = arg != NNZ
    flat = arg[valid]
    kk = torch.arange(K, device="cuda").expand_as(arg)[valid]
    gsel = G[valid]
    # grad_value[e] = sum over (r, k) with arg == e of B[col[e], k] * G[r, k]: float64 scatter, then compare a sample
    ref_v = torch.zeros(NNZ, dtype=torch.float64, device="cuda")
    ref_v.index_add_(0, flat, B[col[flat], kk].double() * gsel.double())
    sc_v = torch.zeros(NNZ, dtype=torch.float64, device="cuda")
    sc_v.index_add_(0, flat, (B[col[flat], kk].double() * gsel.double()).abs())
    assert bool(((gv1.double() - ref_v).abs() <= 1e-5 * sc_v + 1e-30).all())
    # grad_mat on a column sample
    cols = torch.unique(torch.randint(0, N, (3000,), generator=g, device="cuda"))
    pick = torch.isin(col[flat], cols)
    slot = torch.searchsorted(cols, col[flat][pick])
    t = val[flat][pick].double() * gsel[pick].double()
    ref_m = torch.zeros(cols.numel() * K, dtype=torch.float64, device="cuda")
    ref_m.index_add_(0, slot * K + kk[pick], t)
    sc_m = torch.zeros(cols.numel() * K, dtype=torch.float64, device="cuda")
    sc_m.index_add_(0, slot * K + kk[pick], t.abs())
    assert bool(((gm1[cols].double().flatten() - ref_m).abs() <= 1e-5 * sc_m + 1e-30).all())
    # linearity in grad_out (the routing does not depend on it)
    G2 = torch.randn(M, K, generator=g, device="cuda")
    gv2, gm2 = grads(G2)
    gv12, gm12 = grads(G + 2 * G2)
    assert bool(((gv12 - (gv1 + 2 * gv2)).abs() <= 1e-4 * (gv1.abs() + 2 * gv2.abs()) + 1e-4).all())
    assert bool(((gm12 - (gm1 + 2 * gm2)).abs().amax(1) <= 1e-4 * (gm1.abs() + 2 * gm2.abs()).amax(1) + 1e-4).all())


# ---- a power-law graph at scale: R-MAT scale 21 as generated (19.5 M entries, longest row 41 677) ----

@pytest.fixture(scope="module")
def rmat():
    from paddle_sparse_amd import SparseTensor, coalesce, ops

    scale, n = 21, 20_000_000
    size = 1 << scale
    g = torch.Generator(device="cuda").manual_seed(4)
    row = torch.zeros(n, dtype=torch.int64, device="cuda")
    col = torch.zeros(n, dtype=torch.int64, device="cuda")
    for bit in range(scale):
        r = torch.rand(n, generator=g, device="cuda")
        row |= (r >= 0.76).to(torch.int64) << bit
        col |= (((r >= 0.57) & (r < 0.76)) | (r >= 0.95)).to(torch.int64) << bit
    index, val = coalesce(torch.stack([row, col]), torch.randn(n, generator=g, device="cuda"), size, size)
    row, col = index[0].contiguous(), index[1].contiguous()
    B = torch.randn(size, K, generator=g, device="cuda")
    G = torch.randn(size, K, generator=g, device="cuda")
    return dict(row=row, col=col, val=val, rowptr=ops.ind2ptr(row, size), B=B, G=G, size=size)


@pytest.mark.parametrize("reduce", ["max", "min"])
def test_power_law_minmax_training_step_equals_the_int64_route(rmat, reduce):
    """The tensor surface on R-MAT 21 — edge-range forward with the hub-row copy, the two-byte
    row-local arg_out, the one-pass backward with hub-row copies and XCD mixing — gives, bit
    for bit, what the plain route gives: row-wave forward with the int64 arg_out, the same
    backward pass fed with arg_out and one-byte tags, no copies."""
    from paddle_sparse_amd import SparseStorage, SparseTensor, ops

    size, row, col, rowptr = rmat["size"], rmat["row"], rmat["col"], rmat["rowptr"]
    v = rmat["val"].clone().requires_grad_(True)
    Bt = rmat["B"].clone().requires_grad_(True)
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(size, size), is_sorted=True, trust_data=True)
    out = a.matmul(Bt, reduce)
    out.backward(rmat["G"])
    st = a.storage
    assert st._spmm_algo() == "edge_ranges" and st._hot_columns() is not None
    assert 128 < st._longest_row() <= 65_536 and st._csc_view()._hot_columns() is not None
    plain_out, arg = ops._spmm(reduce, rowptr, col, rmat["val"], rmat["B"], algo="row_waves")
    assert torch.equal(out.detach(), plain_out)
    ref = SparseStorage(row=row, rowptr=rowptr, col=col, value=rmat["val"], sparse_sizes=(size, size), is_sorted=True,
                        trust_data=True)
    gv, gm = ops.spmm_minmax_bw_csc(rowptr, ref.colptr(), ref._row_in_csc_order(), ref.csr2csc(), ref._csc_edge_tags(1),
                                    rmat["val"], rmat["B"], rmat["G"], arg, csc2csr=ref.csc2csr())
    assert torch.equal(Bt.grad, gm) and torch.equal(v.grad, gv)
    # the winners really win: out[r, k] == value[arg] * B[col[arg], k] on a sample of rows, hub row 0 included
    rows = torch.cat([torch.tensor([0, 1, 2, 4], device="cuda"), torch.randint(0, size, (2000,), device="cuda")])
    live = arg[rows] < col.numel()
    e = arg[rows].clamp(max=col.numel() - 1)
    want = rmat["val"][e] * torch.gather(rmat["B"][col[e].flatten()].view(*e.shape, K), 2,
                                        torch.arange(K, device="cuda").expand(*e.shape).unsqueeze(-1)).squeeze(-1)
    assert torch.equal(torch.where(live, want, torch.zeros_like(want)), plain_out[rows])


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_power_law_sum_training_step_properties(rmat, reduce):
    """sum / mean with trained values on R-MAT 21 through the surface (one CSC pass with hub-row
    copies): grad_mat columns and grad_value entries against float64 recomputation on samples,
    and the same bits from the pass without copies."""
    from paddle_sparse_amd import SparseTensor, ops

    size, row, col, rowptr = rmat["size"], rmat["row"], rmat["col"], rmat["rowptr"]
    v = rmat["val"].clone().requires_grad_(True)
    Bt = rmat["B"].clone().requires_grad_(True)
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(size, size), is_sorted=True, trust_data=True)
    a.matmul(Bt, reduce).backward(rmat["G"])
    st = a.storage
    scale = (1.0 / st.rowcount().clamp(min=1).float()) if reduce == "mean" else None
    gv, gm = ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order(), st.csr2csc(), rmat["val"], rmat["B"], rmat["G"], True,
                                 csc2csr=st.csc2csr(), row_scale=scale)
    assert torch.equal(Bt.grad, gm) and torch.equal(v.grad, gv)
    # float64 on a sample of entries (grad_value) and of columns (grad_mat), hubs included
    deg = st.rowcount().clamp(min=1).double()
    e = torch.cat([torch.arange(0, 2000, device="cuda"), torch.randint(0, col.numel(), (20_000,), device="cuda")])
    w = (1.0 / deg[row[e]]) if reduce == "mean" else torch.ones(e.numel(), dtype=torch.float64, device="cuda")
    want_v = (rmat["B"][col[e]].double() * rmat["G"][row[e]].double()).sum(1) * w
    mag_v = (rmat["B"][col[e]].double().abs() * rmat["G"][row[e]].double().abs()).sum(1) * w
    assert bool(((v.grad[e].double() - want_v).abs() <= 1e-5 * mag_v + 1e-30).all())
    cols = torch.cat([torch.tensor([0, 1, 2, 3, 4, 8], device="cuda"), torch.randint(0, size, (300,), device="cuda")])
    colptr, r_csc, perm = st.colptr(), st._row_in_csc_order(), st.csr2csc()
    for c in cols.tolist():
        b, en = int(colptr[c]), int(colptr[c + 1])
        rr = r_csc[b:en]
        ww = rmat["val"][perm[b:en]].double() * ((1.0 / deg[rr]) if reduce == "mean" else 1.0)
        want = (ww[:, None] * rmat["G"][rr].double()).sum(0)
        mag = (ww.abs()[:, None] * rmat["G"][rr].double().abs()).sum(0)
        assert bool(((Bt.grad[c].double() - want).abs() <= 1e-5 * mag + 1e-30).all()), c


# ---- BASELINE config 4 WHOLE on one GPU: 16 M x 16 M, 160 M entries, F = 256 (about 52 GB of HBM) ----

def test_config4_whole_matrix_through_the_sharding_functions():
    """The one matrix of BASELINE config 4 assembled on ONE MI355X and pushed through exactly what the 8 ranks of the
    real job run: `partition_rows_by_nnz(rowptr, 8)` -> per block `shard_csr` + the block's planned local kernel
    (`RowShard.storage()` + `spmm_planned`) into out[r0:r1].  The blocks together must give, bit for bit, the
    unsharded `ops.spmm_sum`; a row sample is recomputed in float64; the nnz shares differ by under 1 %.  (What a
    multi-GPU run adds to this is the exchange of B only — covered by the gloo tests and the RCCL world-1 test.)"""
    from paddle_sparse_amd import ops
    from paddle_sparse_amd.distributed import partition_rows_by_nnz, shard_csr
    from paddle_sparse_amd.matmul import spmm_planned

    Mg = Ng = 16_000_000
    nnz, F, world = 160_000_000, 256, 8
    g = torch.Generator(device="cuda").manual_seed(3)
    row = torch.sort(torch.randint(0, Mg, (nnz,), generator=g, device="cuda"))[0]
    rowptr = ops.ind2ptr(row, Mg)
    del row
    col = torch.randint(0, Ng, (nnz,), generator=g, device="cuda")
    val = torch.randn(nnz, generator=g, device="cuda")
    B = torch.randn(Ng, F, generator=g, device="cuda")
    bounds = partition_rows_by_nnz(rowptr, world)
    assert bounds[0] == 0 and bounds[-1] == Mg and len(bounds) == world + 1
    shares = [int(rowptr[bounds[r + 1]] - rowptr[bounds[r]]) for r in range(world)]
    assert sum(shares) == nnz and max(shares) - min(shares) <= 0.01 * (nnz / world)

    out = torch.empty(Mg, F, device="cuda")
    for r in range(world):
        shard = shard_csr(rowptr, col, val, Ng, bounds, r)
        assert shard.num_rows == bounds[r + 1] - bounds[r] and shard.nnz == shares[r]
        spmm_planned(shard.storage(), shard.value, B, "sum", out=out[bounds[r]:bounds[r + 1]])
        del shard
    whole = ops.spmm_sum(rowptr, col, val, B)
    assert torch.equal(out, whole)
    del out
    # float64 on a row sample: first / last rows of every block and random rows
    edge_rows = torch.tensor([b for r in range(world) for b in (bounds[r], bounds[r + 1] - 1)], device="cuda")
    rows = torch.unique(torch.cat([edge_rows, torch.randint(0, Mg, (4000,), generator=g, device="cuda")]))
    lo, hi = rowptr[rows], rowptr[rows + 1]
    deg = hi - lo
    e = torch.repeat_interleave(lo, deg) + (torch.arange(int(deg.sum()), device="cuda")
                                            - torch.repeat_interleave(torch.cumsum(deg, 0) - deg, deg))
    slot = torch.repeat_interleave(torch.arange(rows.numel(), device="cuda"), deg)
    terms = val[e, None].double() * B[col[e]].double()
    ref = torch.zeros(rows.numel(), F, dtype=torch.float64, device="cuda").index_add_(0, slot, terms)
    mag = torch.zeros(rows.numel(), F, dtype=torch.float64, device="cuda").index_add_(0, slot, terms.abs())
    assert bool(((whole[rows].double() - ref).abs() <= 1e-5 * mag + 1e-30).all())


# ---- R-MAT scale 24 (BASELINE config 5's graph): rows of up to ~140 k entries ----

@pytest.mark.parametrize("reduce", ["max"])
def test_rmat24_minmax_training_step_takes_the_bytes_only_route(reduce):
    """R-MAT scale 24, 100 M generated entries (the graph of BASELINE config 5), F = 128: its hub rows exceed the
    65 535 entries the two-byte row-local arg_out can name, and the training step still allocates no int64 arg_out
    (they are reduced once more in pieces, matmul._huge_piece_winners) — with the bits of the int64 route."""
    import sys

    from paddle_sparse_amd import SparseTensor, coalesce, ops

    mm_mod = sys.modules["paddle_sparse_amd.matmul"]
    scale, n = 24, 100_000_000
    size = 1 << scale
    g = torch.Generator(device="cuda").manual_seed(4)
    row = torch.zeros(n, dtype=torch.int64, device="cuda")
    col = torch.zeros(n, dtype=torch.int64, device="cuda")
    for bit in range(scale):
        r = torch.rand(n, generator=g, device="cuda")
        row |= (r >= 0.76).to(torch.int64) << bit
        col |= (((r >= 0.57) & (r < 0.76)) | (r >= 0.95)).to(torch.int64) << bit
    del r
    index, val = coalesce(torch.stack([row, col]), torch.randn(n, generator=g, device="cuda"), size, size)
    del row, col
    row, col = index[0].contiguous(), index[1].contiguous()
    del index
    rowptr = ops.ind2ptr(row, size)
    B = torch.randn(size, K, generator=g, device="cuda")
    G = torch.randn(size, K, generator=g, device="cuda")

    seen = []
    real = ops._spmm

    def spy(*a, **k):
        seen.append((k.get("want_arg_bytes", False), k.get("want_arg", True)))
        return real(*a, **k)

    res = []
    for pieces in (True, False):
        v = val.clone().requires_grad_(True)
        Bt = B.clone().requires_grad_(True)
        a = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(size, size), is_sorted=True, trust_data=True)
        mm_mod.HUGE_ROW_PIECES = pieces
        ops._spmm = spy
        try:
            out = a.matmul(Bt, reduce)
            out.backward(G)
        finally:
            ops._spmm = real
            mm_mod.HUGE_ROW_PIECES = True
        if pieces:
            assert a.storage._longest_row() > 65_535 and a.storage._huge_rows()["rows"].numel() >= 1
            assert all(s == (2, False) for s in seen), seen
            seen.clear()
        else:
            assert (1, True) in seen
        res.append((out.detach(), Bt.grad, v.grad))
        del a, out, v, Bt
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.parametrize("reduce", ["sum", "max"])
def test_power_law_bf16_training_step_stays_half_width(rmat, reduce):
    """VERDICT r03 #4 at full size: R-MAT 21 with trained values and a bf16 dense operand — the transpose is power-law
    (hub rows of 41 677 entries = long columns of the CSC view), and the step still runs the half-width kernels end to
    end: no fp32 pass is called, no fp32 copy of B / grad_out is made (the half-width pass takes long columns in chunks),
    and both gradients are the fp32 route's on the same rounded operands up to bf16 rounding of grad_mat."""
    from paddle_sparse_amd import SparseTensor, ops

    size, row, col, rowptr = rmat["size"], rmat["row"], rmat["col"], rmat["rowptr"]
    Bh = rmat["B"].to(torch.bfloat16)
    Gh = rmat["G"].to(torch.bfloat16)

    def run(B_, G_):
        v = rmat["val"].clone().requires_grad_(True)
        Bt = B_.clone().requires_grad_(True)
        a = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(size, size), is_sorted=True, trust_data=True)
        a.matmul(Bt, reduce).backward(G_)
        return a, v.grad, Bt.grad

    called = []
    # (value[csr2csc], an nnz-float gather, is not a dense pass)
    names = ("spmm_sum_bw_csc", "spmm_value_bw", "spmm_minmax_bw_csc", "spmm_minmax_bw", "spmm_minmax_bw_eb")
    real = {n: getattr(ops, n) for n in names}
    real_spmm = ops._spmm
    for n in names:
        setattr(ops, n, (lambda n_: lambda *x, **k: called.append(n_) or real[n_](*x, **k))(n))
    ops._spmm = lambda reduce, rp, c, v_, mat, *x, **k: (called.append("fp32 forward") if mat.dtype == torch.float32 else None) or real_spmm(reduce, rp, c, v_, mat, *x, **k)
    try:
        a, gv_h, gm_h = run(Bh, Gh)
    finally:
        for n in names:
            setattr(ops, n, real[n])
        ops._spmm = real_spmm
    assert called == [], called
    assert a.storage._csc_view()._longest_row() > 128 and gm_h.dtype == torch.bfloat16 and gv_h.dtype == torch.float32
    # the same through the allocator: a step on the storage whose plans exist by now peaks at its outputs (out and grad_mat
    # in bf16: 0.54 GB each at [2 M, 128]) plus nnz-sized arrays and the chunk scratch — far below what fp32 copies of
    # mat and grad_out (1.07 GB each) and fp32 gradients would add
    v2 = rmat["val"].clone().requires_grad_(True)
    Bt2 = Bh.clone().requires_grad_(True)
    step_input = a.set_value(v2, layout="coo")
    step_input.matmul(Bt2, reduce).backward(Gh)  # second request: the planned routes get built
    v2.grad = Bt2.grad = None
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    step_input.matmul(Bt2, reduce).backward(Gh)
    torch.cuda.synchronize()
    fp32_dense = size * K * 4
    assert torch.cuda.max_memory_allocated() - base < 1.9 * fp32_dense, (torch.cuda.max_memory_allocated() - base) / fp32_dense
    _, gv_f, gm_f = run(Bh.float(), Gh.float())  # the fp32 route on the same rounded operands
    # sum of absolute terms (for max: an upper bound — only the winners' terms are summed)
    scale_m = SparseTensor(row=row, rowptr=rowptr, col=col, value=rmat["val"].abs(), sparse_sizes=(size, size), is_sorted=True,
                           trust_data=True).t().matmul(Gh.float().abs())
    assert bool(((gm_h.float() - gm_f).abs() <= 2.0 ** -8 * gm_f.abs() + 1e-5 * scale_m + 1e-30).all())
    e = torch.randint(0, col.numel(), (200_000,), device="cuda")
    mag_v = (Bh[col[e]].float().abs() * Gh[row[e]].float().abs()).sum(1)
    assert bool(((gv_h[e] - gv_f[e]).abs() <= 1e-5 * mag_v + 1e-30).all())
    if reduce == "max":  # the same winners: grad_value is non-zero on the same entries
        assert torch.equal(gv_h != 0, gv_f != 0)
