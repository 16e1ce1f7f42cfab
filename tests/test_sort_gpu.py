"""GPU parity: index_sort (bit-exact stable permutation), key construction,
gathers, permutation inverse, unique/segment reduce — vs numpy / the oracle."""
import numpy as np
import pytest
import torch

from oracle import storage_oracle as so

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n,max_value,seed", [
    (1, 1, 0), (1, 10, 1), (5, 1, 2), (63, 7, 3), (64, 300, 4), (2047, 1 << 20, 5),
    (2048, 256, 6), (2049, 257, 7), (10000, 1000 * 1000, 8), (100000, 3, 9),
    (300000, 1 << 40, 10), (1000003, 1 << 48, 11), (5000000, 1 << 33, 12),
    (2500000, (1 << 62) + 12345, 13),
])
def test_index_sort_matches_stable_argsort(n, max_value, seed):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(seed)
    keys = rng.integers(0, max_value, n, dtype=np.int64)
    srt, perm = ops.index_sort(dev(keys), max_value, with_sorted_inputs=True)
    ref = so.index_sort(keys)
    assert np.array_equal(perm.cpu().numpy(), ref)
    assert np.array_equal(srt.cpu().numpy(), keys[ref])
    none, perm2 = ops.index_sort(dev(keys), max_value)
    assert none is None and torch.equal(perm, perm2)


def test_index_sort_heavy_duplicates_and_sorted_input():
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(0)
    keys = np.repeat(rng.integers(0, 50, 40), rng.integers(1, 30000, 40)).astype(np.int64)
    rng.shuffle(keys)
    _, perm = ops.index_sort(dev(keys), 50)
    assert np.array_equal(perm.cpu().numpy(), so.index_sort(keys))
    keys = np.sort(keys)
    _, perm = ops.index_sort(dev(keys), 50)
    assert np.array_equal(perm.cpu().numpy(), np.arange(keys.size))
    # max_value unknown -> derived from the data
    _, perm = ops.index_sort(dev(keys[::-1].copy()))
    assert np.array_equal(perm.cpu().numpy(), so.index_sort(keys[::-1]))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("n", [1, 2047, 4097, 300001, 3000001])
def test_index_sort_scatter_variants(variant, n):
    from paddle_sparse_amd import _lib, ops

    keys = np.random.default_rng(n).integers(0, 1 << 37, n, dtype=np.int64)
    keys[::7] = keys[0]  # duplicates exercise stability
    prev = _lib.load().psa_sort_set_variant(variant)
    try:
        srt, perm = ops.index_sort(dev(keys), 1 << 37, with_sorted_inputs=True)
        # unaligned key pointer (scalar histogram path)
        buf = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        buf[1:] = dev(keys)
        _, perm_u = ops.index_sort(buf[1:], 1 << 37)
    finally:
        _lib.load().psa_sort_set_variant(prev)
    ref = so.index_sort(keys)
    assert np.array_equal(perm.cpu().numpy(), ref) and np.array_equal(srt.cpu().numpy(), keys[ref])
    assert np.array_equal(perm_u.cpu().numpy(), ref)


@pytest.mark.parametrize("n,max_value", [(1, 5), (5000, 1), (100000, 1 << 20), (2000003, 1 << 44)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.int32])
def test_sort_pairs_matches_perm_gather(n, max_value, dtype):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(n)
    keys = rng.integers(0, max_value, n, dtype=np.int64)
    pay = torch.from_numpy(rng.integers(-1000, 1000, n).astype(np.int32)).cuda().to(dtype)
    skeys, spay = ops.sort_pairs(dev(keys), pay, max_value)
    ref = so.index_sort(keys)
    assert np.array_equal(skeys.cpu().numpy(), keys[ref])
    assert torch.equal(spay.cpu(), pay.cpu()[torch.from_numpy(ref)])


@pytest.mark.parametrize("n,first_bit,max_value", [(1, 32, 7), (5000, 32, 1), (100000, 32, 1 << 20), (2000003, 32, 2_000_000),
                                                   (300001, 20, 1 << 30), (300001, 0, 1 << 40), (70000, 40, 1 << 24)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.int32])
def test_sort_pairs_on_a_bit_field(n, first_bit, max_value, dtype):
    """psa_sort_pairs_u32_field: stable order by (key >> first_bit); the low bits
    ride along unsorted.  first_bit = 0 is sort_pairs."""
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(n + first_bit)
    hi = rng.integers(0, max_value, n, dtype=np.int64)
    lo = rng.integers(0, 1 << first_bit, n, dtype=np.int64) if first_bit else np.zeros(n, np.int64)
    keys = (hi << first_bit) | lo
    pay = torch.from_numpy(rng.integers(-1000, 1000, n).astype(np.int32)).cuda().to(dtype)
    skeys, spay = ops.sort_pairs_field(dev(keys), pay, first_bit, max_value)
    ref = np.argsort(hi, kind="stable")
    assert np.array_equal(skeys.cpu().numpy(), keys[ref])
    assert torch.equal(spay.cpu(), pay.cpu()[torch.from_numpy(ref)])
    if first_bit == 0:
        k2, p2 = ops.sort_pairs(dev(keys), pay, max_value)
        assert torch.equal(k2, skeys) and torch.equal(p2, spay)
    with pytest.raises(Exception):
        ops.sort_pairs_field(dev(keys), pay, 60, 1 << 10)


def test_single_sweep_lookback_under_uneven_load():
    """The production sort hands digit counts between workgroups inside one
    launch (decoupled look-back).  Exercise it with a second stream keeping the
    chip busy, odd sizes, sorted and half-constant inputs; every element must
    match torch.sort(stable=True)."""
    from paddle_sparse_amd import ops

    g = torch.Generator(device="cuda").manual_seed(0)
    M = 200_000
    row = torch.sort(torch.randint(0, M, (2_000_000,), generator=g, device="cuda"))[0]
    rowptr = ops.ind2ptr(row, M)
    col = torch.randint(0, M, (2_000_000,), generator=g, device="cuda")
    B = torch.randn(M, 64, device="cuda")
    side = torch.cuda.Stream()
    for rep in range(3):
        for n in (8191, 8193, 100_003, 2_500_000, 9_000_001):
            for bits in (8, 9, 33, 48):
                keys = torch.randint(0, 1 << bits, (n,), generator=g, device="cuda")
                if rep == 1:
                    keys = torch.sort(keys)[0]
                if rep == 2:
                    keys[: n // 2] = keys[0]
                with torch.cuda.stream(side):
                    for _ in range(2):
                        ops.spmm_sum(rowptr, col, None, B)
                srt, perm, status = ops.index_sort_checked(keys, 1 << bits)
                ts, tp = torch.sort(keys, stable=True)
                assert status == 0, "a bounded look-back spin gave up"
                assert torch.equal(srt, ts) and torch.equal(perm, tp), (rep, n, bits)
    torch.cuda.synchronize()


def test_index_sort_empty():
    from paddle_sparse_amd import ops

    srt, perm = ops.index_sort(torch.empty(0, dtype=torch.int64, device="cuda"), 10, True)
    assert perm.numel() == 0 and srt.numel() == 0


def test_make_keys_and_flag():
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(1)
    a, b = rng.integers(0, 1 << 24, 100000), rng.integers(0, 1 << 24, 100000)
    keys, flag = ops.make_keys(dev(a), dev(b), 1 << 24, check_sorted=True)
    assert np.array_equal(keys.cpu().numpy(), a * (1 << 24) + b)
    assert int(flag.item()) == 1
    order = np.lexsort((b, a))
    keys, flag = ops.make_keys(dev(a[order]), dev(b[order]), 1 << 24, check_sorted=True)
    assert int(flag.item()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int32, torch.int64, torch.float16, torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("tail", [(), (1,), (2,), (3,), (4, 5), (64,)])
def test_gather_rows(dtype, tail):
    from paddle_sparse_amd import ops

    g = torch.Generator(device="cuda").manual_seed(0)
    src = (torch.rand((1000,) + tail, generator=g, device="cuda") * 100).to(dtype)
    perm = torch.randperm(1000, generator=g, device="cuda")
    assert torch.equal(ops.gather_rows(src, perm), src[perm])
    idx = torch.randint(0, 1000, (3333,), generator=g, device="cuda")
    assert torch.equal(ops.gather_rows(src, idx), src[idx])


def test_invert_permutation():
    from paddle_sparse_amd import ops

    perm = torch.randperm(123457, device="cuda")
    inv = ops.invert_permutation(perm)
    assert torch.equal(inv[perm], torch.arange(123457, device="cuda"))
    # same answer as the reference's "sort the permutation" (storage.py:444-445)
    assert np.array_equal(inv.cpu().numpy(), so.index_sort(perm.cpu().numpy()))


@pytest.mark.parametrize("n,distinct,seed", [(1, 1, 0), (64, 5, 1), (2048, 2048, 2), (2049, 10, 3), (100000, 30000, 4), (700000, 650000, 5)])
def test_unique_sorted(n, distinct, seed):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(seed)
    N = 1000
    pool = np.sort(rng.choice(10**7, distinct, replace=False))
    keys = np.sort(rng.choice(pool, n)).astype(np.int64)
    count, ptr, row, col = ops.unique_sorted(dev(keys), N)
    mask = np.concatenate([[True], keys[1:] != keys[:-1]])
    assert count == mask.sum()
    assert np.array_equal(ptr.cpu().numpy(), np.concatenate([np.nonzero(mask)[0], [n]]))
    assert np.array_equal(row.cpu().numpy(), keys[mask] // N)
    assert np.array_equal(col.cpu().numpy(), keys[mask] % N)


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("npdtype,tdtype", [(np.float32, torch.float32), (np.float64, torch.float64), (np.int32, torch.int32), (np.int64, torch.int64)])
@pytest.mark.parametrize("tail", [(), (2,), (5,)])
def test_segment_csr_vs_oracle(reduce, npdtype, tdtype, tail):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(7)
    n, nseg = 5000, 700
    src = rng.integers(-40, 40, (n,) + tail).astype(npdtype)  # exact in every dtype
    cuts = np.sort(rng.integers(0, n + 1, nseg - 1))
    indptr = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    got = ops.segment_csr(dev(src), dev(indptr), reduce).cpu().numpy()
    ref = so.segment_csr(src, indptr, reduce)
    if reduce == "mean" and npdtype in (np.float32, np.float64):
        np.testing.assert_allclose(got, ref, rtol=1e-6)
    else:
        assert np.array_equal(got, ref)
    # gathered form: reduce src[perm] without materialising it
    perm = rng.permutation(n).astype(np.int64)
    got = ops.segment_csr(dev(src), dev(indptr), reduce, perm=dev(perm)).cpu().numpy()
    ref = so.segment_csr(src[perm], indptr, reduce)
    if reduce == "mean" and npdtype in (np.float32, np.float64):
        np.testing.assert_allclose(got, ref, rtol=1e-6)
    else:
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("tdtype", [torch.float16, torch.bfloat16])
def test_segment_csr_half(tdtype):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(3)
    src = rng.integers(-8, 8, (3000, 2)).astype(np.float32)
    indptr = np.concatenate([[0], np.sort(rng.integers(0, 3001, 400)), [3000]]).astype(np.int64)
    for reduce in ("sum", "min", "max"):
        got = ops.segment_csr(dev(src).to(tdtype), dev(indptr), reduce).float().cpu().numpy()
        assert np.array_equal(got, so.segment_csr(src, indptr, reduce))


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_segment_csr_long_segments_wave_kernel(reduce):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(5)
    n, nseg = 200000, 300
    src = rng.integers(-5, 5, n).astype(np.float32)
    cuts = np.sort(rng.integers(0, n + 1, nseg - 1))
    cuts[10] = cuts[9]  # an empty segment
    indptr = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    got = ops.segment_csr(dev(src), dev(indptr), reduce).cpu().numpy()
    ref = so.segment_csr_fast(src, indptr, reduce)
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0)


def test_count_read_brings_back_the_sorts_lookback_diagnostic():
    """ops.unique_sorted(after=scratch): the one host read of the run count also carries the
    fault word of the sort that produced the keys (psa_unique_count_after_sort) — 0 on every
    real run; a poisoned workspace stands in for a look-back that gave up."""
    from paddle_sparse_amd import ops
    from paddle_sparse_amd._lib import HipCoreError

    g = torch.Generator(device="cuda").manual_seed(5)
    n, bound = 1_500_000, 3000 * 3000
    keys = torch.randint(0, bound, (n,), generator=g, device="cuda")
    out, perm, scratch = ops.index_sort(keys, bound, with_sorted_inputs=True, keep_scratch=True)
    plain = ops.unique_sorted(out, 3000)
    checked = ops.unique_sorted(out, 3000, after=scratch)
    assert plain[0] == checked[0] == int(torch.unique(keys).numel())
    assert all(torch.equal(a, b) for a, b in zip(plain[1:], checked[1:]))
    pay = torch.arange(n, dtype=torch.int32, device="cuda")
    out2, pay2, scratch2 = ops.sort_pairs(keys, pay, bound, keep_scratch=True)
    assert torch.equal(out2, out) and torch.equal(pay2.long(), perm)
    assert ops.unique_sorted(out2, 3000, after=scratch2)[0] == plain[0]
    scratch.ws.fill_(255)
    with pytest.raises(HipCoreError, match="gave up"):
        ops.unique_sorted(out, 3000, after=scratch)


@pytest.mark.parametrize("n,bits", [((1 << 25) + 5, 44), (40_000_003, 17)])
def test_huge_inputs_take_the_16384_key_tiles(n, bits):
    """From 2^25 keys the single-sweep passes run with 1024 x 16 tiles (one workgroup per CU): the
    stable permutation against torch's stable sort, keys and 4-byte payloads; a ragged last tile."""
    from paddle_sparse_amd import ops

    g = torch.Generator(device="cuda").manual_seed(n % 1000)
    keys = torch.randint(0, 1 << bits, (n,), generator=g, device="cuda")
    out, perm, scratch = ops.index_sort(keys, 1 << bits, with_sorted_inputs=True, keep_scratch=True)
    ref = torch.sort(keys, stable=True)
    assert torch.equal(out, ref.values) and torch.equal(perm, ref.indices)
    assert ops.unique_sorted(out, 1 << 20, want_ptr=False, want_rowcol=False, after=scratch)[0] == int(torch.unique(keys).numel())
    del ref
    pay = torch.arange(n, dtype=torch.int32, device="cuda")
    out2, pay2 = ops.sort_pairs(keys, pay, 1 << bits)
    assert torch.equal(out2, out) and torch.equal(pay2.long(), perm)


def test_a_sort_whose_lookback_gave_up_is_never_handed_back_as_an_order():
    """VERDICT r02 #5.  The spin limit of the look-back is set to 0 (psa_sort_set_spin_limit): every
    wait that does not succeed at its first poll gives up, which across ~500 concurrently running tiles
    is certain.  Then (a) the fault word reads non-zero, (b) the last pass stored -1 over the outputs of
    the tiles that ran after the fault instead of positions, (c) the callers whose result no host read
    follows — the SparseStorage constructor sort and csr2csc — raise HipCoreError, as do the checked
    forms of the ops and the count read behind a sort.  With the limit restored the same calls succeed
    and give the stable order (the reference's argsort cannot return garbage: storage.py:164-169)."""
    from paddle_sparse_amd import SparseStorage, _lib, ops
    from paddle_sparse_amd._lib import HipCoreError

    n, M = 4_000_000, 1 << 20
    rng = np.random.default_rng(5)
    row = rng.integers(0, M, n, dtype=np.int64)
    col = rng.integers(0, M, n, dtype=np.int64)
    keys = dev(row * M + col)
    lib = _lib.load()
    prev = lib.psa_sort_set_spin_limit(0)
    try:
        srt, perm, status = ops.index_sort_checked(keys, M * M)
        assert status == 1
        assert int((perm < 0).sum()) > 0 and int((srt < 0).sum()) > 0  # marked, not plausible
        with pytest.raises(HipCoreError, match="gave up"):
            ops.index_sort(keys, M * M, check=True)
        with pytest.raises(HipCoreError, match="gave up"):
            ops.sort_pairs(keys, torch.zeros(n, device="cuda"), M * M, check=True)
        with pytest.raises(HipCoreError, match="gave up"):  # constructor sort, fp32 value rides the sort
            SparseStorage(row=dev(row), col=dev(col), value=torch.ones(n, device="cuda"), sparse_sizes=(M, M))
        with pytest.raises(HipCoreError, match="gave up"):  # constructor sort, permutation form
            SparseStorage(row=dev(row), col=dev(col), value=torch.ones(n, 2, device="cuda"), sparse_sizes=(M, M))
        lib.psa_sort_set_spin_limit(-1)
        st = SparseStorage(row=dev(row), col=dev(col), sparse_sizes=(M, M))
        lib.psa_sort_set_spin_limit(0)
        with pytest.raises(HipCoreError, match="gave up"):
            st.csr2csc()
        assert st._csr2csc is None  # nothing half-built is kept
    finally:
        lib.psa_sort_set_spin_limit(prev)
    assert lib.psa_sort_set_spin_limit(-1) == prev == 1 << 22
    perm_ok = st.csr2csc().cpu().numpy()
    col_sorted = st.col().cpu().numpy()
    assert np.array_equal(perm_ok, np.argsort(col_sorted, kind="stable"))


@pytest.mark.parametrize("n", [1, 5, 32_767, 32_768, 32_769, 100_003, 3 * 32_768 + 5, 2_500_000])
@pytest.mark.parametrize("dtype", [torch.float32, torch.int32])
def test_planned_permutation_equals_the_gather(n, dtype):
    """ops.permute_apply(src, plan of perm) == src[perm] bit for bit: random permutations, sizes
    around the 32 768-element tile, a last partial tile / block, and structured permutations
    (identity, reversal, a transpose-like stride) whose blocks receive from one or from every tile."""
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(n)
    perms = [rng.permutation(n), np.arange(n), np.arange(n)[::-1].copy()]
    if n > 1000:
        w = 257
        idx = np.arange(n)
        perms.append(np.argsort((idx % w) * (n // w + 1) + idx // w, kind="stable"))
    src = dev(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)).view(dtype)
    for perm in perms:
        perm = perm.astype(np.int64)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(n)
        plan = ops.permute_plan(dev(inv))
        got = ops.permute_apply(src, plan)
        assert torch.equal(got.view(torch.int32), src.view(torch.int32)[dev(perm)])
    with pytest.raises(ValueError):
        ops.permute_apply(torch.zeros(n + 1, device="cuda"), plan)


def test_storage_plans_route_values_between_csr_and_csc_order():
    """SparseStorage._permute_plan: "to_csc" = value[csr2csc], "to_csr" = its inverse; None below
    ops.PERMUTE_PLAN_FROM entries.  The one-pass backward gives the same bits with the plan as with
    the gather through csc2csr."""
    from paddle_sparse_amd import SparseStorage, ops

    M, N, nnz, K = 300_000, 200_000, 1_200_000, 32
    rng = np.random.default_rng(3)
    row = np.sort(rng.integers(0, M, nnz)).astype(np.int64)
    col = rng.integers(0, N, nnz).astype(np.int64)
    val = torch.randn(nnz, device="cuda")
    st = SparseStorage(row=dev(row), col=dev(col), value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    to_csc, to_csr = st._permute_plan("to_csc", force=True), st._permute_plan("to_csr", force=True)
    assert to_csc is not None and st._permute_plan("to_csc") is to_csc
    v_csc = ops.permute_apply(val, to_csc)
    assert torch.equal(v_csc, val[st.csr2csc()])
    assert torch.equal(ops.permute_apply(v_csc, to_csr), val)
    B, G = torch.randn(N, K, device="cuda"), torch.randn(M, K, device="cuda")
    args = (st.colptr(), st._row_in_csc_order(), st.csr2csc(), val, B, G, True)
    gv0, gm0 = ops.spmm_sum_bw_csc(*args, csc2csr=st.csc2csr())
    gv1, gm1 = ops.spmm_sum_bw_csc(*args, csc2csr=st.csc2csr(), to_csr_plan=to_csr)
    assert torch.equal(gv0, gv1) and torch.equal(gm0, gm1)
    small = SparseStorage(row=dev(row[:1000]), col=dev(col[:1000]), sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    assert small._permute_plan("to_csr", force=True) is None
    # a one-off use does not pay for a plan: the first request answers None, the second builds it
    again = SparseStorage(row=dev(row), col=dev(col), value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    assert again._permute_plan("to_csr") is None and again._permute_plan("to_csr") is not None
