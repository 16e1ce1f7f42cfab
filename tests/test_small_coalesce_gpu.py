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
