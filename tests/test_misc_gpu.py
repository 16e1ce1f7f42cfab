"""GPU parity of the small index kernels (bincount, count2ptr, scatter) and
the SpMM backward entry points called directly through the C-ABI."""
import numpy as np
import pytest
import torch

import oracle
from oracle import storage_oracle as so
from util import random_csr, skewed_csr

pytestmark = pytest.mark.gpu


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n,size,seed", [(0, 5, 0), (1, 1, 1), (1000, 7, 2), (100000, 30000, 3), (3000000, 2000000, 4)])
def test_bincount_and_count2ptr(n, size, seed):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(seed)
    index = rng.integers(0, size, n, dtype=np.int64)
    cnt = ops.bincount(dev(index) if n else torch.empty(0, dtype=torch.int64, device="cuda"), size)
    ref = np.bincount(index, minlength=size).astype(np.int64)
    assert np.array_equal(cnt.cpu().numpy(), ref)
    ptr = ops.count2ptr(cnt).cpu().numpy()
    assert np.array_equal(ptr, np.concatenate([[0], np.cumsum(ref)]))


@pytest.mark.parametrize("n,div,hi_max,seed", [(0, 7, 5, 0), (1, 1, 1, 1), (100_000, 1000, 1000, 2),
                                                (300_000, 3, 1 << 40, 3), (300_000, (1 << 33) + 5, 1 << 29, 4),
                                                (1_000_000, 16_777_216, 16_777_216, 5)])
def test_make_keys_split_keys_round_trip(n, div, hi_max, seed):
    """split_keys undoes make_keys bit for bit (32-bit and 64-bit division paths)."""
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(seed)
    a = rng.integers(0, hi_max, n, dtype=np.int64)
    b = rng.integers(0, div, n, dtype=np.int64)
    a_d = dev(a) if n else torch.empty(0, dtype=torch.int64, device="cuda")
    b_d = dev(b) if n else torch.empty(0, dtype=torch.int64, device="cuda")
    keys, _ = ops.make_keys(a_d, b_d, div)
    assert np.array_equal(keys.cpu().numpy(), a * div + b)
    hi, lo = ops.split_keys(keys, div)
    assert np.array_equal(hi.cpu().numpy(), a) and np.array_equal(lo.cpu().numpy(), b)
    hi, lo = ops.split_keys(keys, div, want_lo=False)
    assert lo is None and np.array_equal(hi.cpu().numpy(), a)
    hi, lo = ops.split_keys(keys, div, want_hi=False)
    assert hi is None and np.array_equal(lo.cpu().numpy(), b)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("npdtype", [np.float32, np.float64, np.int32, np.int64])
@pytest.mark.parametrize("tail", [(), (3,), (64,)])
def test_scatter_vs_oracle(reduce, npdtype, tail):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(11)
    n, size = 4000, 300
    src = rng.integers(-20, 20, (n,) + tail).astype(npdtype)  # exact sums
    index = rng.integers(0, size - 10, n, dtype=np.int64)      # last rows untouched
    got = ops.scatter(dev(src), dev(index), size, reduce).cpu().numpy()
    ref = so.scatter(src, index, size, reduce)
    if reduce == "mean" and npdtype in (np.float32, np.float64):
        np.testing.assert_allclose(got, ref, rtol=1e-6)
    else:
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("K", [1, 3, 4, 16, 64, 128, 256, 300, 520])
def test_spmm_value_bw(reduce, K):
    from paddle_sparse_amd import ops

    row, rowptr, col, val = skewed_csr(300, 250, seed=K, long_rows=(0, 299), long_deg=200)
    rng = np.random.default_rng(K)
    B = rng.standard_normal((250, K)).astype(np.float32)
    G = rng.standard_normal((300, K)).astype(np.float32)
    got = ops.spmm_value_bw(dev(row), dev(rowptr), dev(col), dev(B), dev(G), reduce).cpu().numpy()
    ref = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
    scale = np.abs(B[col] * G[row]).sum(axis=1) + 1e-30
    deg = np.maximum(rowptr[1:] - rowptr[:-1], 1)[row] if reduce == "mean" else 1
    assert np.all(np.abs(got - ref) <= 1e-5 * scale / deg)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("K", [8, 64, 128, 300])
def test_spmm_value_bw_long_rows(reduce, K):
    """Rows above 128 edges go to chunk waves (disjoint out[e] ranges)."""
    from paddle_sparse_amd import ops

    row, rowptr, col, val = skewed_csr(500, 400, seed=K, long_rows=(0, 250, 499), long_deg=4000)
    rng = np.random.default_rng(K)
    B = rng.standard_normal((400, K)).astype(np.float32)
    G = rng.standard_normal((500, K)).astype(np.float32)
    got = ops.spmm_value_bw(None, dev(rowptr), dev(col), dev(B), dev(G), reduce).cpu().numpy()
    ref = oracle.spmm_value_bw(reduce, row, rowptr, col, B, G)
    scale = np.abs(B[col] * G[row]).sum(axis=1) + 1e-30
    deg = np.maximum(rowptr[1:] - rowptr[:-1], 1)[row] if reduce == "mean" else 1
    assert np.all(np.abs(got - ref) <= 1e-5 * scale / deg)


@pytest.mark.parametrize("mean", [False, True])
@pytest.mark.parametrize("has_value", [False, True])
def test_transposed_spmm_is_grad_mat(mean, has_value):
    """gB = A^T gOut evaluated as psa_spmm over the CSC view."""
    from paddle_sparse_amd import ops

    M, N, K = 500, 350, 64
    rng = np.random.default_rng(2)
    key = np.unique(rng.integers(0, M * N, 7000))
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32) if has_value else None
    G = rng.standard_normal((M, K)).astype(np.float32)
    rowptr = oracle.ind2ptr(row, M)
    st = so.Storage(row, col, val, (M, N))
    perm = st.csr2csc()
    w = ops.transpose_weights(dev(val), dev(perm), dev(row[perm]), dev(rowptr), mean)
    gB = ops.spmm_sum(dev(st.colptr()), dev(row[perm]), w, dev(G)).cpu().numpy()
    ref = oracle.spmm_mat_bw("mean" if mean else "sum", row, rowptr, col, val, G, N)
    np.testing.assert_allclose(gB, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("has_value", [True, False])
@pytest.mark.parametrize("K", [4, 16, 32, 64, 100, 128, 256])
@pytest.mark.parametrize("graph", ["uniform", "hubs"])
def test_minmax_bw_over_csc(reduce, has_value, K, graph):
    """The atomic-free min/max backward (one CSC gather pass over byte-compressed
    arg_out, both gradients) against the oracle: short rows take the byte test,
    rows of > 255 edges the exact one, columns of > 128 edges the chunked path."""
    from paddle_sparse_amd import SparseStorage, ops

    if graph == "hubs":  # rows 0 / 7 have 700 edges; few columns, so columns are long too
        row, rowptr, col, val = skewed_csr(600, 40, seed=3 + K, long_rows=(0, 7, 599), long_deg=700)
        M, N = 600, 40
    else:
        row, rowptr, col, val = random_csr(3000, 2500, 30_000, 4 + K)
        M, N = 3000, 2500
    if not has_value:
        val = None
    rng = np.random.default_rng(2)
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    _, arg = oracle.spmm(reduce, rowptr, col, val, B)
    ref_v, ref_m = oracle.spmm_minmax_bw(col, val, B, G, arg)
    st = SparseStorage(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(M, N), is_sorted=True)
    csr2csc = st.csr2csc()

    def run(want_value=True):
        return ops.spmm_minmax_bw_csc(st.rowptr(), st.colptr(), st._row_in_csc_order(), csr2csc,
                                      st._csc_edge_tags(), st.value(), dev(B), dev(G), dev(arg),
                                      want_value=want_value)

    gv, gm = run()
    live = arg < col.size
    rr, kk = np.nonzero(live)
    ee = arg[rr, kk]
    scale_m = np.zeros((N, K), np.float32)  # sum of |terms| per output element
    w = np.ones(col.size, np.float32) if val is None else np.abs(val)
    np.add.at(scale_m, (col[ee], kk), w[ee] * np.abs(G[rr, kk]))
    assert np.all(np.abs(gm.cpu().numpy() - ref_m) <= 1e-5 * scale_m + 1e-30)
    scale_v = np.zeros(col.size, np.float32)
    np.add.at(scale_v, ee, np.abs(B[col[ee], kk] * G[rr, kk]))
    assert np.all(np.abs(gv.cpu().numpy() - ref_v) <= 1e-5 * scale_v + 1e-30)
    assert ops.minmax_bw_csc_supported(K)
    # two launches give the same bits (no atomics anywhere on this path)
    gv2, gm2 = run()
    assert torch.equal(gm, gm2) and torch.equal(gv, gv2)
    none, gm3 = run(want_value=False)
    assert none is None and torch.equal(gm, gm3)


@pytest.mark.parametrize("has_value", [True, False])
@pytest.mark.parametrize("K", [4, 32, 64, 100, 128, 256])
@pytest.mark.parametrize("graph", ["uniform", "hubs"])
def test_sum_bw_one_csc_pass(has_value, K, graph):
    """psa_spmm_sum_bw_csc (both sum gradients from one pass over the CSC view)
    against the oracle's spmm_value_bw / spmm_mat_bw."""
    from paddle_sparse_amd import SparseStorage, ops

    if graph == "hubs":
        row, rowptr, col, val = skewed_csr(600, 40, seed=5 + K, long_rows=(0, 7, 599), long_deg=700)
        M, N = 600, 40
    else:
        row, rowptr, col, val = random_csr(3000, 2500, 30_000, 6 + K)
        M, N = 3000, 2500
    if not has_value:
        val = None
    rng = np.random.default_rng(3)
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    ref_m = oracle.spmm_mat_bw("sum", row, rowptr, col, val, G, N)
    ref_v = oracle.spmm_value_bw("sum", row, rowptr, col, B, G)
    st = SparseStorage(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(M, N), is_sorted=True)
    csr2csc = st.csr2csc()
    gv, gm = ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order(), csr2csc, st.value(), dev(B), dev(G), True,
                                 csc2csr=st.csc2csr())
    w = np.ones(col.size, np.float32) if val is None else np.abs(val)
    scale_m = np.zeros((N, K), np.float32)
    np.add.at(scale_m, col, w[:, None] * np.abs(G[row]))
    assert np.all(np.abs(gm.cpu().numpy() - ref_m) <= 1e-5 * scale_m + 1e-30)
    scale_v = (np.abs(B[col]) * np.abs(G[row])).sum(1)
    assert np.all(np.abs(gv.cpu().numpy() - ref_v) <= 1e-5 * scale_v + 1e-30)
    # mean: 1 / deg(row) folded into both gradients per edge
    deg = np.maximum(rowptr[1:] - rowptr[:-1], 1).astype(np.float32)
    gv_mean, gm_mean = ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order(), csr2csc, st.value(), dev(B), dev(G),
                                           True, csc2csr=st.csc2csr(), row_scale=dev(1.0 / deg))
    ref_mm = oracle.spmm_mat_bw("mean", row, rowptr, col, val, G, N)
    ref_mv = oracle.spmm_value_bw("mean", row, rowptr, col, B, G)
    np.testing.assert_allclose(gm_mean.cpu().numpy(), ref_mm, rtol=1e-4, atol=1e-5 * float(scale_m.max() + 1))
    np.testing.assert_allclose(gv_mean.cpu().numpy(), ref_mv, rtol=1e-4, atol=1e-5 * float(scale_v.max() + 1))
    # without the inverse permutation handed in, the op builds it itself; grad_mat alone skips the dots
    gv2, gm2 = ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order(), csr2csc, st.value(), dev(B), dev(G), True)
    none, gm3 = ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order(), csr2csc, st.value(), None, dev(G), False)
    assert torch.equal(gv, gv2) and torch.equal(gm, gm2) and none is None and torch.equal(gm, gm3)


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [8, 32, 64, 100, 128, 256])
def test_forward_leaves_arg_out_as_row_local_bytes(reduce, K):
    """psa_spmm(arg_bytes=...): the byte form of arg_out the one-pass backward
    reads, written by the forward itself (fused-roles kernel and its long-row
    combine) or by the compress pass behind the other kernels."""
    from paddle_sparse_amd import SparseStorage, ops

    # rows of 0, a few, 200 and 700 edges (> 128: chunked, and flagged in the byte form)
    row, rowptr, col, val = skewed_csr(900, 300, seed=K, long_rows=(0, 450), long_deg=700)
    row2, rowptr2, col2, val2 = skewed_csr(900, 300, seed=K + 1, long_rows=(3, 899), long_deg=200)
    for rp, c, v in ((rowptr, col, val), (rowptr2, col2, val2)):
        B = np.random.default_rng(K).standard_normal((300, K)).astype(np.float32)
        out, arg, ab = ops._spmm(reduce, dev(rp), dev(c), dev(v), dev(B), want_arg_bytes=True)
        ref_out, ref_arg = oracle.spmm(reduce, rp, c, v, B)
        assert np.array_equal(arg.cpu().numpy(), ref_arg)
        deg = rp[1:] - rp[:-1]
        live = deg > 0
        want = (((ref_arg - rp[:-1, None]) & 127) | np.where(deg > 128, 0x80, 0)[:, None]).astype(np.uint8)
        assert ab.dtype == torch.uint8 and np.array_equal(ab.cpu().numpy()[live], want[live])
        # the backward gives the same bits with the bytes handed in or derived inside
        st = SparseStorage(rowptr=dev(rp), col=dev(c), value=dev(v), sparse_sizes=(900, 300), is_sorted=True)
        G = torch.randn(900, K, device="cuda")
        args = (st.rowptr(), st.colptr(), st._row_in_csc_order(), st.csr2csc(), st._csc_edge_tags(), st.value(),
                dev(B), G, arg)
        gv1, gm1 = ops.spmm_minmax_bw_csc(*args, csc2csr=st.csc2csr(), arg_bytes=ab)
        gv2, gm2 = ops.spmm_minmax_bw_csc(*args, csc2csr=st.csc2csr())
        assert torch.equal(gv1, gv2) and torch.equal(gm1, gm2)


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [4, 32, 64, 100, 128, 256, 130, 384])
def test_minmax_forward_without_arg_out(reduce, K):
    """psa_spmm(arg_out=NULL): `out` (and the byte form, where the kernel writes
    it itself) without the int64 arg_out — same bits as the full call, on every
    kernel (multirow, fused roles + long-row combine, row kernel, scalar lanes)."""
    from paddle_sparse_amd import SparseStorage, ops
    from paddle_sparse_amd._lib import HipCoreError

    for long_deg in (100, 200, 700):  # short rows only; chunked rows past the byte form's exact reach (128)
        row, rowptr, col, val = skewed_csr(900, 300, seed=K + long_deg, long_rows=(0, 450), long_deg=long_deg)
        B = np.random.default_rng(K).standard_normal((300, K)).astype(np.float32)
        full = ops._spmm(reduce, dev(rowptr), dev(col), dev(val), dev(B), want_arg_bytes=True)
        out_only = ops._spmm(reduce, dev(rowptr), dev(col), dev(val), dev(B), want_arg=False)
        assert out_only[1] is None and torch.equal(out_only[0], full[0])
        if not ops.minmax_bw_csc_supported(K):
            # no kernel writes the bytes itself for this K tile: the wrapper does not ask for them ...
            res = ops._spmm(reduce, dev(rowptr), dev(col), dev(val), dev(B), want_arg_bytes=True, want_arg=False)
            assert res[1] is None and res[2] is None and torch.equal(res[0], full[0])
            continue
        out, arg, ab = ops._spmm(reduce, dev(rowptr), dev(col), dev(val), dev(B), want_arg_bytes=True, want_arg=False)
        assert arg is None and torch.equal(out, full[0]) and torch.equal(ab, full[2])
        # the backward served by the bytes alone: exact when no row is longer than 255
        st = SparseStorage(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(900, 300), is_sorted=True)
        assert st._longest_row() == long_deg
        G = torch.randn(900, K, device="cuda")
        head = (st.rowptr(), st.colptr(), st._row_in_csc_order(), st.csr2csc(), st._csc_edge_tags(), st.value(),
                dev(B), G)
        gv_ref, gm_ref = ops.spmm_minmax_bw_csc(*head, full[1], csc2csr=st.csc2csr(), arg_bytes=full[2])
        gv, gm = ops.spmm_minmax_bw_csc(*head, None, csc2csr=st.csc2csr(), arg_bytes=ab)
        if long_deg <= ops.ARG_BYTES_EXACT_ROW:
            assert torch.equal(gv, gv_ref) and torch.equal(gm, gm_ref)
        else:  # documented: entries of rows past 128 count as no hit without arg_out (and nothing faults)
            short = torch.from_numpy(((rowptr[1:] - rowptr[:-1]) <= 128)[row]).cuda()
            assert torch.equal(gv[short], gv_ref[short])
    if ops.minmax_bw_csc_supported(K):
        with pytest.raises(ValueError, match="arg_out or arg_bytes"):
            ops.spmm_minmax_bw_csc(*head, None, csc2csr=st.csc2csr())
    with pytest.raises(HipCoreError, match="arg_bytes without arg_out"):
        # straight through the C-ABI: bytes asked for, no arg_out, a K tile whose kernel cannot write them
        from paddle_sparse_amd import _lib
        Bw = torch.randn(300, 260, device="cuda")
        o = torch.empty(900, 260, device="cuda")
        byt = torch.empty(900, 260, dtype=torch.uint8, device="cuda")
        _lib.check(_lib.load().psa_spmm(_lib.REDUCE_ID[reduce], dev(rowptr).data_ptr(), dev(col).data_ptr(), None,
                                        Bw.data_ptr(), 900, 300, 260, col.size, o.data_ptr(), None, byt.data_ptr(),
                                        None, 0, torch.cuda.current_stream().cuda_stream))


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [4, 32, 64, 128, 256])
@pytest.mark.parametrize("algo", ["row_waves", "edge_ranges"])
def test_two_byte_row_local_arg(reduce, K, algo):
    """want_arg_bytes=2: arg_out as (index in the row) & 0xffff, exact for rows of up to 65 536
    entries, written by every forward kernel family; the one-pass backward served by it alone
    gives the bits of the one-byte + arg_out route and the oracle's gradients."""
    from paddle_sparse_amd import SparseStorage, ops

    M, N = 900, 300
    row, rowptr, col, val = skewed_csr(M, N, seed=K, long_rows=(0, 450, 899), long_deg=3000)
    row2, rowptr2, col2, val2 = skewed_csr(M, N, seed=K + 1, long_rows=(3, 600), long_deg=200)
    for r, rp, c, v in ((row, rowptr, col, val), (row2, rowptr2, col2, val2)):
        B = np.random.default_rng(K).standard_normal((N, K)).astype(np.float32)
        out, arg, words = ops._spmm(reduce, dev(rp), dev(c), dev(v), dev(B), want_arg_bytes=2, row=dev(r), algo=algo)
        ref_out, ref_arg = oracle.spmm(reduce, rp, c, v, B)
        assert np.array_equal(arg.cpu().numpy(), ref_arg) and np.array_equal(out.cpu().numpy(), ref_out)
        live = (rp[1:] - rp[:-1]) > 0
        want = ((ref_arg - rp[:-1, None]) & 0xffff).astype(np.uint16).view(np.int16)
        assert words.dtype == torch.int16 and np.array_equal(words.cpu().numpy()[live], want[live])
        only = ops._spmm(reduce, dev(rp), dev(c), dev(v), dev(B), want_arg_bytes=2, want_arg=False, row=dev(r), algo=algo)
        assert only[1] is None and torch.equal(only[0], out) and torch.equal(only[2][dev(live)], words[dev(live)])
        st = SparseStorage(rowptr=dev(rp), col=dev(c), value=dev(v), sparse_sizes=(M, N), is_sorted=True)
        G = torch.randn(M, K, device="cuda")
        head = (st.rowptr(), st.colptr(), st._row_in_csc_order(), st.csr2csc())
        tail = (st.value(), dev(B), G)
        gv2, gm2 = ops.spmm_minmax_bw_csc(*head, st._csc_edge_tags(2), *tail, None, csc2csr=st.csc2csr(), arg_bytes=words)
        gv1, gm1 = ops.spmm_minmax_bw_csc(*head, st._csc_edge_tags(1), *tail, arg, csc2csr=st.csc2csr())
        assert torch.equal(gv1, gv2) and torch.equal(gm1, gm2)
        # ... and derived inside the call from arg_out (no forward bytes at hand)
        gv3, gm3 = ops.spmm_minmax_bw_csc(*head, st._csc_edge_tags(2), *tail, arg, csc2csr=st.csc2csr())
        assert torch.equal(gv1, gv3) and torch.equal(gm1, gm3)
        ref_v, ref_m = oracle.spmm_minmax_bw(c, v, B, G.cpu().numpy(), ref_arg)
        np.testing.assert_allclose(gm2.cpu().numpy(), ref_m, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(gv2.cpu().numpy(), ref_v, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("K", [16, 64, 128, 256])
def test_csc_backward_reads_hub_rows_from_compact_copies(reduce, K):
    """hot_ids of spmm_sum_bw_csc / spmm_minmax_bw_csc: rows of grad (and of arg_bytes) that the
    CSC view refers to as M + position are read from compact copies — same terms in the same
    order, so both gradients keep their bits."""
    from paddle_sparse_amd import SparseStorage, ops

    M, N = 900, 300
    row, rowptr, col, val = skewed_csr(M, N, seed=K, long_rows=(0, 450, 899), long_deg=2500)
    st = SparseStorage(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(M, N), is_sorted=True)
    csr2csc = st.csr2csc()
    row_csc = st._row_in_csc_order()
    hot = torch.tensor([899, 0, 17, 450, 3, 444], device="cuda")
    slot = torch.full((M,), -1, dtype=torch.int64, device="cuda")
    slot[hot] = torch.arange(hot.numel(), device="cuda")
    redirected = torch.where(slot[row_csc] >= 0, M + slot[row_csc], row_csc)
    B = torch.randn(N, K, device="cuda")
    G = torch.randn(M, K, device="cuda")
    if reduce in ("sum", "mean"):
        scale = (1.0 / st.rowcount().clamp(min=1).float()) if reduce == "mean" else None
        plain = ops.spmm_sum_bw_csc(st.colptr(), row_csc, csr2csc, st.value(), B, G, True, csc2csr=st.csc2csr(), row_scale=scale)
        copy = ops.spmm_sum_bw_csc(st.colptr(), redirected, csr2csc, st.value(), B, G, True, csc2csr=st.csc2csr(),
                                   row_scale=scale, hot_ids=hot)
    else:
        out, _, words = ops._spmm(reduce, st.rowptr(), st.col(), st.value(), B, want_arg_bytes=2, want_arg=False)
        tags = st._csc_edge_tags(2)
        plain = ops.spmm_minmax_bw_csc(st.rowptr(), st.colptr(), row_csc, csr2csc, tags, st.value(), B, G, None,
                                       csc2csr=st.csc2csr(), arg_bytes=words)
        copy = ops.spmm_minmax_bw_csc(st.rowptr(), st.colptr(), redirected, csr2csc, tags, st.value(), B, G, None,
                                      csc2csr=st.csc2csr(), arg_bytes=words, hot_ids=hot)
        with pytest.raises(ValueError, match="exact arg_bytes"):
            ops.spmm_minmax_bw_csc(st.rowptr(), st.colptr(), redirected, csr2csc, tags, st.value(), B, G,
                                   torch.zeros(M, K, dtype=torch.int64, device="cuda"), arg_bytes=words, hot_ids=hot)
    assert torch.equal(plain[0], copy[0]) and torch.equal(plain[1], copy[1])


@pytest.mark.parametrize("reduce", ["min", "max"])
def test_autograd_minmax_skips_arg_out_when_the_bytes_suffice(reduce):
    """matmul.py: short rows + grad of the dense operand -> the forward keeps the
    byte form only; long rows or value-only gradients keep arg_out; no gradient
    at all keeps neither.  Gradients are the same bits as with arg_out kept."""
    import sys

    from paddle_sparse_amd import SparseTensor, ops

    mm_mod = sys.modules["paddle_sparse_amd.matmul"]  # the package exports a function of the same name

    seen = []
    real = ops._spmm

    def spy(*a, **k):
        seen.append((k.get("want_arg_bytes", False), k.get("want_arg", True)))
        return real(*a, **k)

    # rows up to 128 entries: one byte per element; up to 65 535: two; beyond: two as well — the rows concerned are
    # reduced once more in pieces of at most 65 535 entries (matmul._huge_piece_winners: the last _spmm call seen)
    for long_deg, expect in ((100, (1, False)), (200, (2, False)), (66_000, (2, False)), (140_000, (2, False))):
        row, rowptr, col, val = skewed_csr(700, 300, seed=long_deg, long_rows=(5,), long_deg=long_deg)
        B = torch.randn(300, 64, device="cuda")
        grads = []
        for patched in (False, True):
            A = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=dev(val).requires_grad_(True),
                             sparse_sizes=(700, 300), is_sorted=True)
            Bg = B.clone().requires_grad_(True)
            if patched:  # reference run: force the full arg_out (and the int64 route for rows above 65 535 entries)
                ops._spmm = lambda *a, **k: real(*a, **{**k, "want_arg": True})
                mm_mod.HUGE_ROW_PIECES = False
            else:
                ops._spmm = spy
            try:
                out = mm_mod.spmm_sparse(A, Bg, reduce)
                out.backward(torch.randn(out.shape, generator=torch.Generator().manual_seed(long_deg)).cuda())
            finally:
                ops._spmm = real
                mm_mod.HUGE_ROW_PIECES = True
            grads.append((A.storage.value().grad.clone(), Bg.grad.clone(), out.detach()))
        assert seen[-1] == expect
        for x, y in zip(*grads):
            assert torch.equal(x, y)
    # value gradient only -> the atomics backward needs arg_out; no gradient -> nothing is kept
    A = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=dev(val).requires_grad_(True), sparse_sizes=(700, 300),
                     is_sorted=True)
    ops._spmm = spy
    try:
        mm_mod.spmm_sparse(A, B, reduce).sum().backward()
        assert seen[-1] == (False, True) and A.storage.value().grad is not None
        with torch.no_grad():
            ref = real(reduce, dev(rowptr), dev(col), dev(val), B)[0]
            got = mm_mod.spmm_sparse(A, B, reduce)
        assert seen[-1] == (False, False) and torch.equal(got, ref)
    finally:
        ops._spmm = real


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [68, 128, 256])
@pytest.mark.parametrize("algo", ["row_waves", "edge_ranges"])
def test_tiny_matrix_minmax_backward_wrt_the_dense_operand(reduce, K, algo):
    """at most 128 entries (no row can be long) and 64 < K <= 256 — the fused-roles
    kernel is not taken, so the byte form of arg_out must come from arg_out itself.  The call used
    to raise HipCoreError (bytes asked for, neither a kernel that writes them nor an arg_out)."""
    import sys

    from paddle_sparse_amd import SparseTensor, ops

    mm_mod = sys.modules["paddle_sparse_amd.matmul"]
    rng = np.random.default_rng(K)
    M = N = 8
    deg = np.array([5, 6, 5, 4, 5, 5, 5, 5])  # 40 entries, every row at least 3
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([rng.choice(N, d, replace=False) for d in deg]).astype(np.int64)
    val = rng.standard_normal(40).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    G = rng.standard_normal((M, K)).astype(np.float32)
    # the entry point itself: bytes without arg_out
    res = ops._spmm(reduce, dev(rowptr), dev(col), dev(val), dev(B), want_arg_bytes=1, want_arg=False, algo=algo)
    ref, ref_arg = oracle.spmm(reduce, rowptr, col, val, B)
    assert np.array_equal(res[0].cpu().numpy(), ref)
    assert np.array_equal(res[2].cpu().numpy().astype(np.int64), ref_arg - rowptr[:-1, None])
    # ... and through autograd with a fixed adjacency
    A = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(M, N), is_sorted=True)
    A.storage._spmm_algo_memo = algo
    Bg = dev(B).requires_grad_(True)
    mm_mod.spmm_sparse(A, Bg, reduce).backward(dev(G))
    gm = np.zeros((N, K), np.float64)
    for i in range(M):
        for k in range(K):
            e = ref_arg[i, k]
            gm[col[e], k] += float(val[e]) * float(G[i, k])
    np.testing.assert_allclose(Bg.grad.cpu().numpy(), gm, rtol=1e-5, atol=1e-6)


def test_minmax_bw_over_csc_rejects_unaligned_k():
    from paddle_sparse_amd import ops
    from paddle_sparse_amd._lib import HipCoreError

    assert not ops.minmax_bw_csc_supported(130) and not ops.minmax_bw_csc_supported(512)
    z = torch.zeros(3, dtype=torch.int64, device="cuda")
    with pytest.raises(HipCoreError, match="K % 4"):
        ops.spmm_minmax_bw_csc(z, z, z[:0], z[:0], torch.zeros(0, dtype=torch.uint8, device="cuda"), None,
                               torch.zeros(2, 6, device="cuda"), torch.zeros(2, 6, device="cuda"),
                               torch.zeros(2, 6, dtype=torch.int64, device="cuda"))


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [5, 48, 100, 128, 130, 256, 300])
def test_spmm_minmax_bw(reduce, K):
    from paddle_sparse_amd import ops

    row, rowptr, col, val = skewed_csr(400, 300, seed=8 + K, long_rows=(0, 200), long_deg=300)
    rng = np.random.default_rng(1)
    B = rng.standard_normal((300, K)).astype(np.float32)
    G = rng.standard_normal((400, K)).astype(np.float32)
    _, arg = oracle.spmm(reduce, rowptr, col, val, B)
    gv, gm = ops.spmm_minmax_bw(dev(col), dev(val), dev(B), dev(G), dev(arg))
    rv, rm = oracle.spmm_minmax_bw(col, val, B, G, arg)
    np.testing.assert_allclose(gv.cpu().numpy(), rv, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gm.cpu().numpy(), rm, rtol=1e-4, atol=1e-5)
    only_v, none = ops.spmm_minmax_bw(dev(col), dev(val), dev(B), dev(G), dev(arg), want_mat=False)
    # float atomics: summation order differs between launches
    assert none is None and torch.allclose(only_v, gv, rtol=1e-4, atol=1e-5)


def test_value_in_csc_order_is_kept_until_either_side_changes():
    """value[csr2csc] (dim-0 reduction, csc(), t()) is memoised on the storage; an in-place
    write to the values or to the memoised copy must be seen."""
    from paddle_sparse_amd import SparseTensor

    rng = np.random.default_rng(8)
    M, N = 200, 150
    key = np.unique(rng.integers(0, M * N, 3000))
    row, col = key // N, key % N
    val = rng.standard_normal(key.size).astype(np.float32)
    v = torch.from_numpy(val).cuda()
    a = SparseTensor(row=torch.from_numpy(row).cuda(), col=torch.from_numpy(col).cuda(), value=v, sparse_sizes=(M, N))
    a.storage.csr2csc()

    def colsum(values):
        out = np.zeros(N, np.float64)
        np.add.at(out, col, values.astype(np.float64))
        return out

    first = a.sum(0)
    memo = a.storage._value_csc_memo[2]
    assert a.storage._value_in_csc_order() is memo and a.csc()[2] is memo
    np.testing.assert_allclose(first.cpu().numpy(), colsum(val), rtol=1e-5, atol=1e-5)
    v.mul_(3.0)  # the values change in place: same tensor, new version
    np.testing.assert_allclose(a.sum(0).cpu().numpy(), colsum(3 * val), rtol=1e-5, atol=1e-5)
    assert a.storage._value_csc_memo[2] is not memo
    t = a.t()
    t.storage.value().add_(1.0)  # the transposed tensor owns what was the memo: written in place
    np.testing.assert_allclose(a.sum(0).cpu().numpy(), colsum(3 * val), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("K", [4, 16, 32, 64, 128, 256])  # the K tiles whose forward leaves the row-local arg_out
@pytest.mark.parametrize("width", [1, 2])
@pytest.mark.parametrize("has_value", [True, False])
def test_minmax_grad_mat_by_edge_ranges_over_the_csc_view(reduce, K, width, has_value):
    """psa_spmm_minmax_bw_eb: grad wrt the dense operand for a fixed adjacency — the edge-range kernels
    over the CSC view, masked by the forward's row-local arg_out — against the oracle and the row-wave
    pass; with and without the view's COO ids, with hub-row copies; two launches give the same bits."""
    from paddle_sparse_amd import SparseStorage, ops

    M, N = 900, 300
    long_deg = 100 if width == 1 else 2500  # width 1 is exact up to 128 entries per row
    row, rowptr, col, val = skewed_csr(M, N, seed=K + width, long_rows=(0, 450, 899), long_deg=long_deg)
    if not has_value:
        val = None
    st = SparseStorage(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(M, N), is_sorted=True)
    csr2csc, row_csc = st.csr2csc(), st._row_in_csc_order()
    B = torch.randn(N, K, device="cuda")
    G = torch.randn(M, K, device="cuda")
    _, _, words = ops._spmm(reduce, st.rowptr(), st.col(), st.value(), B, want_arg_bytes=width, want_arg=False)
    tags = st._csc_edge_tags(width)
    w = None if val is None else ops.transpose_weights(st.value(), csr2csc, None, None, False)
    view_row = st._csc_view().row()
    got = ops.spmm_minmax_bw_eb(st.colptr(), view_row, row_csc, tags, w, G, words)
    _, ref_arg = oracle.spmm(reduce, rowptr, col, val, B.cpu().numpy())
    _, ref_m = oracle.spmm_minmax_bw(col, val, B.cpu().numpy(), G.cpu().numpy(), ref_arg, want_value=False)
    live = ref_arg < col.size
    rr, kk = np.nonzero(live)
    ee = ref_arg[rr, kk]
    scale = np.zeros((N, K), np.float32)
    wabs = np.ones(col.size, np.float32) if val is None else np.abs(val)
    np.add.at(scale, (col[ee], kk), wabs[ee] * np.abs(G.cpu().numpy()[rr, kk]))
    assert np.all(np.abs(got.cpu().numpy() - ref_m) <= 1e-5 * scale + 1e-30)
    none, waves = ops.spmm_minmax_bw_csc(st.rowptr(), st.colptr(), row_csc, csr2csc, tags, st.value(), B, G, None,
                                         want_value=False, arg_bytes=words)
    assert none is None and np.all(np.abs(got.cpu().numpy() - waves.cpu().numpy()) <= 2e-5 * scale + 1e-30)
    assert torch.equal(got, ops.spmm_minmax_bw_eb(st.colptr(), view_row, row_csc, tags, w, G, words))
    assert torch.equal(got, ops.spmm_minmax_bw_eb(st.colptr(), None, row_csc, tags, w, G, words))  # ids derived from colptr
    hot = torch.tensor([899, 0, 17, 450, 3, 444], device="cuda")
    slot = torch.full((M,), -1, dtype=torch.int64, device="cuda")
    slot[hot] = torch.arange(hot.numel(), device="cuda")
    redirected = torch.where(slot[row_csc] >= 0, M + slot[row_csc], row_csc)
    assert torch.equal(got, ops.spmm_minmax_bw_eb(st.colptr(), view_row, redirected, tags, w, G, words, hot_ids=hot))


@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_autograd_steps_agree_before_and_after_the_planned_routes_exist(reduce, dtype):
    """A storage builds its planned CSR <-> CSC routes on the SECOND request (storage._permute_plan): the first
    backward of a training loop gathers through csr2csc / csc2csr and reads row_scale inside the pass, later ones
    stream the weights, fold the mean scale into them and bring grad_value back along the plan.  Same gradients:
    grad_value bit for bit for sum / max in fp32 (same terms, same order), within 1e-5 * sum|terms| for mean (the
    scale multiplies at another point) and for the half-width passes."""
    from paddle_sparse_amd import SparseTensor, ops

    M, N, nnz, K = 250_000, 200_000, 1_300_000, 32
    row, rowptr, col, val = random_csr(M, N, nnz, seed=11, sort_cols=True)
    keep = np.concatenate([[True], (row[1:] != row[:-1]) | (col[1:] != col[:-1])])
    row, col, val = row[keep], col[keep], val[keep]
    assert col.size >= ops.PERMUTE_PLAN_FROM
    v = dev(val).requires_grad_()
    B = torch.randn(N, K, device="cuda").to(dtype).requires_grad_()
    G = torch.randn(M, K, device="cuda").to(dtype)
    a = SparseTensor(row=dev(row), col=dev(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    grads = []
    for step in range(3):
        v.grad = B.grad = None
        a.matmul(B, reduce).backward(G)
        grads.append((v.grad.clone(), B.grad.clone()))
        if step == 0:
            assert not any(isinstance(k, str) for k in a.storage._perm_plans)  # asked once, not built yet
    built = set(k for k in a.storage._perm_plans if isinstance(k, str))
    assert "to_csr" in built and built <= {"to_csr", "to_csc"}  # (to_csc is not asked for again while value[csr2csc] is memoised)
    (gv0, gm0), (gv2, gm2) = grads[0], grads[2]
    assert torch.equal(grads[1][0], gv2) and torch.equal(grads[1][1], gm2)  # steps 2 and 3 run the same code
    if dtype == torch.float32 and reduce != "mean":
        assert torch.equal(gv0, gv2) and torch.equal(gm0, gm2)
    else:
        Bf, Gf = B.detach().float(), G.float()
        scale_v = ops.spmm_value_bw(None, a.storage.rowptr(), a.storage.col(), Bf.abs(), Gf.abs(), "sum") + 1e-30
        assert bool(((gv0 - gv2).abs() <= 1e-5 * scale_v).all())
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 0.0
        scale_m = SparseTensor(row=dev(row), col=dev(col), value=dev(np.abs(val)), sparse_sizes=(M, N),
                               is_sorted=True).t().matmul(Gf.abs()) + 1e-30
        assert bool(((gm0.float() - gm2.float()).abs() <= 1e-5 * scale_m + 2 * eps * gm2.float().abs()).all())


@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
@pytest.mark.parametrize("K", [16, 64, 96, 128, 192])
@pytest.mark.parametrize("long_deg", [20, 300])
def test_tensor_surface_skips_the_long_row_launches_only_when_no_row_is_long(reduce, K, long_deg):
    """SparseTensor.matmul hands `no_long_rows` to the op when its storage knows that no row has more than 128
    entries: for K <= 128 the call then brings no long-row workspace (no list / chunk / combine launches).  Same
    bits as the raw op, which always brings it; with a long row present the hint is off and nothing changes."""
    from paddle_sparse_amd import SparseTensor, ops

    row, rowptr, col, val = skewed_csr(3000, 2000, seed=K + long_deg, long_rows=(5, 1500), long_deg=long_deg)
    B = torch.randn(2000, K, device="cuda")
    a = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=dev(val), sparse_sizes=(3000, 2000), is_sorted=True)
    a.storage._spmm_algo_memo = "row_waves"  # (skewed_csr has many empty rows: keep the family fixed for the comparison)
    a.storage._longest_row()
    seen = []
    real = ops._spmm
    ops._spmm = lambda *x, **k: seen.append(k.get("no_long_rows", False)) or real(*x, **k)
    try:
        with torch.no_grad():
            got = a.matmul(B, reduce)
    finally:
        ops._spmm = real
    assert seen == [long_deg <= 128]
    want = real(reduce, dev(rowptr), dev(col), dev(val), B, want_arg=False, algo="row_waves")[0]
    assert torch.equal(got, want)


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("case", ["trained_values", "fixed_adjacency", "no_values_ties", "trained_unit_values_ties"])
def test_rows_above_65535_entries_keep_the_bytes_only_route(reduce, case):
    """A 200 000-entry row, a 70 000-entry row (two pieces, the second short) and one of exactly 65 535 + 65 535
    entries among tiny rows (a power-law shape: edge-range forward, edge-range CSC view): the training step
    allocates no int64 arg_out — the long rows are reduced once more in pieces of at most 65 535 entries whose
    two-byte winners are exact (matmul._huge_piece_winners, SparseStorage._huge_backward_plan) — and gives, bit
    for bit, the gradients of the int64 route.  no_values_ties: value None and a dense operand of small integers,
    so most products tie and the winner must be the FIRST edge reaching the extreme."""
    import sys

    from paddle_sparse_amd import SparseTensor, ops

    mm_mod = sys.modules["paddle_sparse_amd.matmul"]
    rng = np.random.default_rng(11)
    M, N, K = 3000, 200_000, 64
    deg = rng.integers(0, 3, M)
    deg[7], deg[1500], deg[2999], deg[40] = 200_000, 70_000, 131_070, 30_000
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = np.minimum((rng.random(nnz) ** 4 * N).astype(np.int64), N - 1)  # most columns hold at most two entries
    ties = case.endswith("ties")
    val = None if case == "no_values_ties" else (np.ones(nnz, np.float32) if ties else rng.standard_normal(nnz).astype(np.float32))
    B = rng.integers(-3, 4, (N, K)).astype(np.float32) if ties else rng.standard_normal((N, K)).astype(np.float32)
    G = torch.randn(M, K, generator=torch.Generator().manual_seed(3)).cuda()
    train = case.startswith("trained")

    seen = []
    real = ops._spmm

    def spy(*a, **k):
        seen.append((k.get("want_arg_bytes", False), k.get("want_arg", True)))
        return real(*a, **k)

    res = []
    for pieces in (True, False):
        v = None if val is None else dev(val).requires_grad_(train)
        A = SparseTensor(rowptr=dev(rowptr), col=dev(col), value=v, sparse_sizes=(M, N), is_sorted=True)
        Bg = dev(B).requires_grad_(True)
        mm_mod.HUGE_ROW_PIECES = pieces
        ops._spmm = spy
        try:
            out = mm_mod.spmm_sparse(A, Bg, reduce)
            out.backward(G)
        finally:
            ops._spmm = real
            mm_mod.HUGE_ROW_PIECES = True
        if pieces:
            st = A.storage
            assert st._spmm_algo() == "edge_ranges" and st._csc_view()._spmm_algo() == "edge_ranges"
            hr = st._huge_rows()
            assert hr["rows"].tolist() == [7, 1500, 2999] and hr["piece_ptr"].tolist() == [0, 4, 6, 8]
            assert hr["rowptr"].tolist()[-1] == 200_000 + 70_000 + 131_070
            assert all(s == (2, False) for s in seen)  # the product and the pieces: two-byte winners, no int64 arg_out
            seen.clear()
        else:
            assert (1, True) in seen  # the reference run did take the int64 route
        res.append((out.detach(), Bg.grad, None if not train else v.grad))
    (o1, gm1, gv1), (o2, gm2, gv2) = res
    assert torch.equal(o1, o2)
    if not train:
        # fixed adjacency on this shape: the edge-range kernels over the CSC view, which add a column's terms in
        # another order than the row-wave pass of the int64 route — same terms (the same winners), fp32 rounding apart
        terms = torch.zeros(N, K, dtype=torch.float64, device="cuda")
        _, arg = real(reduce, dev(rowptr), dev(col), None if val is None else dev(val), dev(B))
        live = arg < nnz
        e = arg[live]
        kk = torch.arange(K, device="cuda").expand_as(arg)[live]
        w = torch.ones(e.numel(), dtype=torch.float64, device="cuda") if val is None else dev(val)[e].double()
        terms.index_put_((dev(col)[e], kk), (w * G[live].double()).abs(), accumulate=True)
        assert bool(((gm1.double() - gm2.double()).abs() <= 1e-6 * terms + 1e-30).all())
    else:
        assert torch.equal(gm1, gm2)
    if train:
        assert torch.equal(gv1, gv2)
