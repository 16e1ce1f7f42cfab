"""GPU parity: ind2ptr / ptr2ind through the C-ABI vs the oracle, bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from util import random_csr, skewed_csr

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_kats(kats):
    from paddle_sparse_amd import ops

    for c in kats["ind2ptr"]["cases"]:
        got = ops.ind2ptr(dev(np.array(c["row"], np.int64)), c["M"])
        assert got.cpu().tolist() == c["rowptr"]
    for c in kats["ptr2ind"]["cases"]:
        got = ops.ptr2ind(dev(np.array(c["rowptr"], np.int64)), c["E"])
        assert got.cpu().tolist() == c["row"]


@pytest.mark.parametrize("M,nnz,seed", [
    (1, 1, 0), (1, 700, 1), (7, 0, 2), (64, 64, 3), (65, 1, 4), (1000, 10000, 5),
    (100000, 300, 6),      # long runs of empty rows -> wave-cooperative path
    (257, 100000, 7),      # many duplicates per row
    (300000, 3000000, 8),
])
def test_random(M, nnz, seed):
    from paddle_sparse_amd import ops

    row, rowptr, _, _ = random_csr(M, 4, nnz, seed)
    got = ops.ind2ptr(dev(row), M).cpu().numpy()
    assert np.array_equal(got, oracle.ind2ptr(row, M))
    # an index that starts 8 bytes into its buffer takes the one-boundary-per-lane kernel
    shifted = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), dev(row)])[1:]
    assert nnz == 0 or shifted.data_ptr() % 16 == 8
    assert np.array_equal(ops.ind2ptr(shifted, M).cpu().numpy(), got)
    back = ops.ptr2ind(dev(rowptr), nnz).cpu().numpy()
    assert np.array_equal(back, oracle.ptr2ind(rowptr, nnz))
    assert np.array_equal(back, row)
    # an output that starts 8 bytes into its buffer takes the 8-byte-store kernel (straight C-ABI call)
    from paddle_sparse_amd import _lib
    buf = torch.full((nnz + 1,), -7, dtype=torch.int64, device="cuda")
    ptr_d = dev(rowptr)
    _lib.check(_lib.load().psa_ptr2ind(ptr_d.data_ptr(), M, nnz, buf[1:].data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
    assert int(buf[0]) == -7 and np.array_equal(buf[1:].cpu().numpy(), row)


def test_first_and_last_rows_only():
    from paddle_sparse_amd import ops

    M = 50000
    row = np.array([0, 0, M - 1, M - 1, M - 1], np.int64)
    got = ops.ind2ptr(dev(row), M).cpu().numpy()
    assert np.array_equal(got, oracle.ind2ptr(row, M))
    row = np.array([M // 2], np.int64)
    got = ops.ind2ptr(dev(row), M).cpu().numpy()
    assert np.array_equal(got, oracle.ind2ptr(row, M))


def test_skewed_rows():
    from paddle_sparse_amd import ops

    row, rowptr, _, _ = skewed_csr(5000, 10, seed=3, long_rows=(0, 63, 64, 4999), long_deg=5000)
    nnz = row.size
    assert np.array_equal(ops.ind2ptr(dev(row), 5000).cpu().numpy(), rowptr)
    assert np.array_equal(ops.ptr2ind(dev(rowptr), nnz).cpu().numpy(), row)


def test_roundtrip_full_size():
    """C3-sized round trip (size-independent property: ptr2ind(ind2ptr(r)) == r)."""
    from paddle_sparse_amd import ops

    M, nnz = 2_000_000, 20_000_000
    g = torch.Generator(device="cuda").manual_seed(0)
    row = torch.sort(torch.randint(0, M, (nnz,), generator=g, device="cuda"))[0]
    rowptr = ops.ind2ptr(row, M)
    assert int(rowptr[0]) == 0 and int(rowptr[-1]) == nnz
    assert bool((rowptr[1:] >= rowptr[:-1]).all())
    assert torch.equal(rowptr, torch.searchsorted(row, torch.arange(M + 1, device="cuda")))
    assert torch.equal(ops.ptr2ind(rowptr, nnz), row)


def test_ptr2ind_blocks_dominated_by_hub_rows():
    """64-row blocks with far more entries than rows take the row-by-row fill (no search):
    odd and even starts, hub rows next to empty and one-entry rows, the last partial block."""
    from paddle_sparse_amd import ops

    rng = np.random.default_rng(3)
    for M in (1, 63, 64, 65, 200):
        deg = rng.integers(0, 3, M)
        deg[rng.integers(0, M, max(M // 20, 1))] = rng.integers(5000, 40000, max(M // 20, 1))
        deg[0] = 4097 + (M % 2)  # an odd start for the second row of block 0
        ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        E = int(ptr[-1])
        got = ops.ptr2ind(torch.from_numpy(ptr).cuda(), E).cpu().numpy()
        assert np.array_equal(got, np.repeat(np.arange(M), deg))
        assert np.array_equal(got, oracle.ptr2ind(ptr, E))
