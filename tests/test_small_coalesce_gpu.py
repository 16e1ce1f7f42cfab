"""GPU: the one-workgroup small-input path (psa_coalesce_small: sort by
(row, col) + run-length structure in one launch) against the stable argsort
and the numpy coalesce oracle, bit for bit — one and several 10240-key tile
steps, 1 to 7 radix passes, duplicates, sorted and constant inputs."""
import numpy as np
import pytest
import torch

from oracle import storage_oracle as so

pytestmark = pytest.mark.gpu


def idx(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.int64).cuda()


@pytest.mark.parametrize("n,M,N,seed", [
    (1, 1, 1, 0), (2, 1, 2, 1), (63, 5, 7, 2), (64, 300, 300, 3), (65, 3, 2, 4), (1000, 1000, 1000, 5),
    (10_000, 1000, 1000, 6),            # BASELINE config 1
    (10_240, 70_000, 90_000, 7),        # exactly one tile step, 5 passes
    (10_241, 2, 3, 8),                  # second tile step holds one key; massive duplication
    (30_000, 1 << 20, 1 << 20, 9),      # three tile steps, 5 passes
    (40_960, 1 << 27, 1 << 27, 10),     # the maximum: four tile steps, 7 passes
    (5000, 1 << 30, 1 << 31, 11),       # 8 passes
])
def test_small_path_equals_stable_argsort_and_unique(n, M, N, seed):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(seed)
    row, col = rng.integers(0, M, n), rng.integers(0, N, n)
    key = row.astype(np.int64) * N + col
    count, ptr, r, c, perm = ops.coalesce_small(idx(row), idx(col), M, N)
    ref_perm = so.index_sort(key)
    assert np.array_equal(perm.cpu().numpy(), ref_perm)
    skey = key[ref_perm]
    heads = np.flatnonzero(np.concatenate([[True], skey[1:] != skey[:-1]]))
    assert count == heads.size
    assert np.array_equal(ptr.cpu().numpy(), np.concatenate([heads, [n]]))
    assert np.array_equal(r.cpu().numpy(), skey[heads] // N) and np.array_equal(c.cpu().numpy(), skey[heads] % N)


def test_small_path_sorted_constant_and_limits():
    from paddle_sparse_amd import ops
    from paddle_sparse_amd._lib import HipCoreError

    n = 20_000
    row = np.sort(np.random.default_rng(1).integers(0, 500, n))
    col = np.zeros(n, np.int64)
    count, ptr, r, c, perm = ops.coalesce_small(idx(row), idx(col), 500, 1)
    assert np.array_equal(perm.cpu().numpy(), np.arange(n))  # stable: sorted input keeps its order
    assert count == np.unique(row).size and torch.equal(r, torch.unique(idx(row)))
    count, ptr, r, c, perm = ops.coalesce_small(idx(np.full(n, 3)), idx(np.full(n, 4)), 10, 10)
    assert count == 1 and ptr.tolist() == [0, n] and r.tolist() == [3] and c.tolist() == [4]
    assert np.array_equal(perm.cpu().numpy(), np.arange(n))
    assert ops.coalesce_small_max() == 40_960
    big = idx(np.zeros(40_961, np.int64))
    with pytest.raises(HipCoreError, match="out of range"):
        ops.coalesce_small(big, big, 4, 4)


@pytest.mark.parametrize("op", ["add", "mean", "min", "max"])
@pytest.mark.parametrize("shape_tail,npdtype", [((), np.float32), ((2,), np.float32), ((), np.int64), ((3,), np.float64)])
def test_coalesce_takes_the_small_path_with_the_same_results(op, shape_tail, npdtype):
    """Up to 10240 entries coalesce() routes through the one-workgroup kernel;
    results equal the numpy oracle (integers and min/max bit for bit)."""
    import paddle_sparse_amd as ps

    rng = np.random.default_rng(4)
    n, M, N = 10_000, 1000, 1000
    row, col = rng.integers(0, M, n) % 60, rng.integers(0, N, n) % 50  # ~3 duplicates per entry
    val = rng.integers(-9, 10, (n,) + shape_tail).astype(npdtype)
    index = np.stack([row, col])
    ref_i, ref_v = so.coalesce(index, val, M, N, op)
    got_i, got_v = ps.coalesce(idx(index), torch.from_numpy(val).cuda(), M, N, op)
    assert np.array_equal(got_i.cpu().numpy(), ref_i) and got_i.is_contiguous()
    if npdtype in (np.float32, np.float64) and op == "mean":
        np.testing.assert_allclose(got_v.cpu().numpy(), ref_v, rtol=1e-6)
    else:
        assert np.array_equal(got_v.cpu().numpy(), ref_v)
    tr_i, tr_v = so.transpose(index, val, M, N)
    gt_i, gt_v = ps.transpose(idx(index), torch.from_numpy(val).cuda(), M, N)
    assert np.array_equal(gt_i.cpu().numpy(), tr_i) and np.array_equal(gt_v.cpu().numpy(), tr_v)


# ---- the two-call chain (psa_coalesce_count / psa_coalesce_write) -------------------------------

@pytest.mark.parametrize("n,M,N,seed", [
    (1, 1, 1, 0), (2, 1, 2, 1), (63, 5, 7, 2), (64, 300, 300, 3), (65, 3, 2, 4), (1000, 1000, 1000, 5),
    (10_000, 1000, 1000, 6),            # BASELINE config 1: one workgroup, sort resident in the LDS
    (10_240, 70_000, 90_000, 7),        # the one-workgroup limit, 5 passes
    (10_241, 2, 3, 8),                  # first size on the multi-launch path; massive duplication
    (5000, 1 << 30, 1 << 31, 11),       # 8 passes in the LDS
    (300_000, 1 << 20, 1 << 20, 9), (1_000_000, 3000, 3000, 10),
])
@pytest.mark.parametrize("read_first", [False, True])
def test_chain_equals_the_oracle(n, M, N, seed, read_first):
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(seed)
    row, col = rng.integers(0, M, n), rng.integers(0, N, n)
    index = np.stack([row, col])
    for val, op in ((None, "add"), (rng.integers(-9, 10, n).astype(np.float32), "add"),
                    (rng.integers(-9, 10, (n, 2)).astype(np.float64), "max"),
                    (rng.integers(-9, 10, n).astype(np.int64), "min"), (rng.integers(-9, 10, n).astype(np.int32), "mean")):
        ref_i, ref_v = so.coalesce(index, val, M, N, op)
        got_i, got_v, was_sorted = ops.coalesce_chain(idx(row), idx(col), None if val is None else torch.from_numpy(val).cuda(),
                                                      M, N, op, read_first=read_first)
        assert got_i.is_contiguous() and np.array_equal(got_i.cpu().numpy(), ref_i)
        assert (got_v is None) == (val is None)
        if val is not None:
            assert np.array_equal(got_v.cpu().numpy(), ref_v)
        key = row.astype(object) * N + col
        assert was_sorted == bool(np.all(key[1:] >= key[:-1]))


@pytest.mark.parametrize("op", ["add", "mean", "min", "max"])
@pytest.mark.parametrize("npdtype", [np.float32, np.int32])
@pytest.mark.parametrize("n,M,N", [(1, 1, 1), (777, 9, 11), (10_000, 1000, 1000), (10_240, 40, 50)])
def test_one_launch_form_equals_the_two_call_chain_and_the_oracle(op, npdtype, n, M, N):
    """psa_coalesce_small_fused (taken by coalesce_chain for <= 10240 entries with 4-byte scalar
    values) against the oracle and against the two-call chain on the same input, bit for bit."""
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(n + M)
    row, col = rng.integers(0, M, n), rng.integers(0, N, n)
    val = rng.integers(-9, 10, n).astype(npdtype)
    if npdtype is np.float32 and op != "mean":
        val = val + rng.random(n).astype(np.float32)   # sums in run order: same order in all three
    ref_i, ref_v = so.coalesce(np.stack([row, col]), val, M, N, op)
    one = ops.coalesce_chain(idx(row), idx(col), torch.from_numpy(val).cuda(), M, N, op)
    two = ops.coalesce_chain(idx(row), idx(col), torch.from_numpy(val).cuda(), M, N, op, read_first=True)
    assert np.array_equal(one[0].cpu().numpy(), ref_i) and one[0].is_contiguous()
    assert torch.equal(one[0], two[0]) and torch.equal(one[1], two[1]) and one[2] == two[2]
    if npdtype is np.float32:
        np.testing.assert_allclose(one[1].cpu().numpy(), ref_v, rtol=1e-5, atol=1e-5)  # order of a float sum
    else:
        assert np.array_equal(one[1].cpu().numpy(), ref_v)


def test_chain_and_functional_forms_reject_indices_outside_the_matrix():
    """The reference asserts row.max() < M and col.max() < N (storage.py:78-91);
    here the key kernels raise a flag that the one host read brings back."""
    import paddle_sparse_amd as ps
    from paddle_sparse_amd import ops

    for n in (100, 50_000, 1_500_000):  # one workgroup, chain, multi-call path
        rng = np.random.default_rng(n)
        row, col = rng.integers(0, 40, n), rng.integers(0, 30, n)
        val = torch.ones(n, device="cuda")
        for bad_row, bad_col in ((40, 0), (0, 30), (-1, 0), (0, -2)):
            r, c = row.copy(), col.copy()
            r[n // 2], c[n // 2] = bad_row, bad_col
            index = idx(np.stack([r, c]))
            with pytest.raises(ops.IndexRangeError):
                ps.coalesce(index, val, 40, 30)
            with pytest.raises(ops.IndexRangeError):
                ps.transpose(index, val, 40, 30)
            with pytest.raises(ops.IndexRangeError):
                ps.spmm(index, val, 40, 30, torch.ones(30, 4, device="cuda"))
        ps.coalesce(idx(np.stack([row, col])), val, 40, 30)  # in range: fine


def test_chain_replays_from_a_hip_graph():
    """Both calls are allocation-free and read nothing on the host: with worst-case
    outputs the whole coalesce is capturable; the status words come back after the replay."""
    from paddle_sparse_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    for n, M, N in ((10_000, 1000, 1000), (200_000, 5000, 7000)):
        row = torch.randint(0, M, (n,), generator=g, device="cuda")
        col = torch.randint(0, N, (n,), generator=g, device="cuda")
        val = torch.randn(n, generator=g, device="cuda")
        ws = torch.empty(lib.psa_coalesce_workspace_bytes(n, M, N), dtype=torch.uint8, device="cuda")
        index = torch.empty(2 * n, dtype=torch.int64, device="cuda")
        out = torch.empty(n, device="cuda")

        def run():
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.psa_coalesce_count(row.data_ptr(), col.data_ptr(), val.data_ptr(), 0, 1, n, M, N,
                                              ws.data_ptr(), ws.numel(), st))
            _lib.check(lib.psa_coalesce_write(val.data_ptr(), 0, 1, n, M, N, 0, -1, ws.data_ptr(), index.data_ptr(),
                                              out.data_ptr(), st))

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
        for trial in range(3):
            row.copy_(torch.randint(0, M, (n,), generator=g, device="cuda"))
            col.copy_(torch.randint(0, N, (n,), generator=g, device="cuda"))
            val.copy_(torch.randn(n, generator=g, device="cuda"))
            graph.replay()
            torch.cuda.synchronize()
            count = int(ws[:8].view(torch.int64).item())
            e_i, e_v, _ = ops.coalesce_chain(row, col, val, M, N, "add")
            assert count == e_i.shape[1]
            assert torch.equal(index[:2 * count].view(2, count), e_i) and torch.equal(out[:count], e_v)


# ---- above the chain's limit (2^20 entries): keys -> pair sort -> run lengths -> ONE launch for index + reduced values ----

@pytest.mark.parametrize("op", ["add", "mean", "min", "max"])
@pytest.mark.parametrize("npdtype", [np.float32, np.int32])
@pytest.mark.parametrize("shape", ["short runs", "long runs", "no duplicates", "sorted input with duplicates"])
def test_coalesce_above_the_chain_limit_equals_the_oracle(op, npdtype, shape):
    """coalesce() of 1.2 M entries (the Python-level path: psa_make_keys_checked, psa_sort_pairs_u32,
    psa_unique_count_after_sort, then psa_unique_write_reduce — index and reduced values from one launch, no ptr array —
    when the runs are short, psa_unique_write + psa_segment_reduce when they are long) against the numpy oracle, bit for
    bit: sums of a run are taken in run order by both forms."""
    import paddle_sparse_amd as ps
    from paddle_sparse_amd import ops

    n = 1_200_000
    rng = np.random.default_rng(len(shape) + len(op))
    if shape == "short runs":
        M, N = 3000, 3500
        row, col = rng.integers(0, M, n), rng.integers(0, N, n)
    elif shape == "long runs":  # 6 000 distinct pairs: 200 entries per run, the wave-per-run reducer
        M, N = 60, 100
        row, col = rng.integers(0, M, n), rng.integers(0, N, n)
    elif shape == "no duplicates":
        M, N = 2000, 1000
        key = rng.permutation(M * N)[:n]
        row, col = key // N, key % N
    else:
        M, N = 3000, 3500
        key = np.sort(rng.integers(0, M * N, n))
        row, col = key // N, key % N
    val = rng.integers(-9, 10, n).astype(npdtype)
    if npdtype is np.float32 and op != "mean":
        val = val + rng.random(n).astype(np.float32)  # fp32 sums in run order: the same order on both sides
    index = np.stack([row, col])
    ref_i, ref_v = so.coalesce(index, val, M, N, op)
    called = []
    real = ops.unique_sorted_reduce
    ops.unique_sorted_reduce = lambda *a, **k: called.append(1) or real(*a, **k)
    try:
        got_i, got_v = ps.coalesce(idx(index), torch.from_numpy(val).cuda(), M, N, op)
    finally:
        ops.unique_sorted_reduce = real
    assert called == [1]
    assert np.array_equal(got_i.cpu().numpy(), ref_i)
    if npdtype is np.float32 and op in ("mean", "add"):
        # at this size the oracle reduces with ufunc.reduceat, whose order inside a run is its own: a tolerance here,
        # the run-order bits in test_unique_sorted_reduce_sums_in_run_order below
        _, S = so.coalesce(index, np.abs(val), M, N, op)  # the fp32 bar of the path: 1e-5 * sum |terms| of the run
        assert np.all(np.abs(got_v.cpu().numpy() - ref_v) <= 1e-5 * S + 1e-30)
    else:
        assert np.array_equal(got_v.cpu().numpy(), ref_v)


@pytest.mark.parametrize("op", ["add", "mean"])
def test_unique_sorted_reduce_sums_in_run_order(op):
    """The one-launch form adds a run's values sequentially, in run order — the bits of the oracle's sequential
    segment_csr (and of psa_segment_reduce's thread-per-segment kernel)."""
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(3)
    n, N = 200_000, 300
    keys = np.sort(rng.integers(0, 60_000, n))  # runs of ~3.3 entries
    val = (rng.integers(-9, 10, n) + rng.random(n)).astype(np.float32)
    count, row, col, out = ops.unique_sorted_reduce(idx(keys), N, torch.from_numpy(val).cuda(), op)
    uniq, start = np.unique(keys, return_index=True)
    ref = so.segment_csr(val, np.concatenate([start, [n]]), op)
    assert count == uniq.size and np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(row.cpu().numpy(), uniq // N) and np.array_equal(col.cpu().numpy(), uniq % N)


def test_unique_sorted_reduce_edge_cases():
    from paddle_sparse_amd import ops

    for keys, N in (([5], 3), ([0, 0, 0, 0], 1), ([1, 1, 2, 2, 2, 9], 4), (list(range(100)), 10)):
        k = np.asarray(keys, np.int64)
        v = np.arange(1, k.size + 1, dtype=np.float32)
        count, row, col, out = ops.unique_sorted_reduce(idx(k), N, torch.from_numpy(v).cuda(), "sum")
        uniq, start = np.unique(k, return_index=True)
        assert count == uniq.size
        assert np.array_equal(row.cpu().numpy(), uniq // N) and np.array_equal(col.cpu().numpy(), uniq % N)
        assert np.array_equal(out.cpu().numpy(), np.add.reduceat(v, start))
    with pytest.raises(ValueError):
        ops.unique_sorted_reduce(idx([1, 2]), 3, torch.zeros(2, dtype=torch.float64, device="cuda"))
