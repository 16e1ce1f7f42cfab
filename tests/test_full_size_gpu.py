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
