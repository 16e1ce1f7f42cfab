"""GPU: spspmm against the README known answer
(README.md:336-353) and bit-exact against the row-by-row CPU oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def random_coo(m, n, nnz, rng, skew=False):
    if skew:  # a few very long rows / very popular columns
        r = np.minimum((rng.pareto(1.2, nnz) * 3).astype(np.int64), m - 1)
        c = np.minimum((rng.pareto(1.2, nnz) * 3).astype(np.int64), n - 1)
        key = np.unique(r * n + c)
    else:
        key = np.unique(rng.integers(0, m * n, nnz))
    return np.stack([key // n, key % n]), rng.standard_normal(key.size).astype(np.float32)


def ref_products(iA, iB, k):
    return int(np.bincount(iB[0], minlength=k)[iA[1]].sum())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int32, torch.int64])
def test_readme_kat(kats, dtype):
    from paddle_sparse_amd import spspmm

    k = kats["spspmm"]
    idx, val = spspmm(dev(np.array(k["indexA"])), dev(np.array(k["valueA"]), dtype),
                      dev(np.array(k["indexB"])), dev(np.array(k["valueB"]), dtype), k["m"], k["k"], k["n"])
    assert idx.tolist() == k["indexC"]
    assert val.dtype == dtype and val.tolist() == k["valueC"]


@pytest.mark.parametrize("m,k,n,nnzA,nnzB,skew", [
    (50, 40, 30, 300, 200, False),
    (3000, 2000, 2500, 40_000, 30_000, False),
    (100_000, 100_000, 100_000, 1_000_000, 1_000_000, False),
    (5000, 5000, 5000, 60_000, 60_000, True),
    (7, 50_000, 9, 9000, 20_000, False),
    (1, 1, 1, 1, 1, False),
])
def test_bit_exact_vs_oracle(m, k, n, nnzA, nnzB, skew):
    """Index AND fp32 values identical: every C entry is summed in the order
    the sequential row-by-row product meets its terms."""
    from paddle_sparse_amd import spspmm

    rng = np.random.default_rng(m + k + n)
    iA, vA = random_coo(m, k, nnzA, rng, skew)
    iB, vB = random_coo(k, n, nnzB, rng, skew)
    ref_idx, ref_val = oracle.spspmm(iA, vA, iB, vB, m, k, n)
    idx, val = spspmm(dev(iA), dev(vA), dev(iB), dev(vB), m, k, n)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    if idx.shape[1] * 32 > ref_products(iA, iB, k):
        assert np.array_equal(val.cpu().numpy(), ref_val)
    else:
        # runs of >= 32 products on average are summed lane-strided (fixed order,
        # not the sequential one): 1e-5 of the sum of |terms|, north_star's bound
        _, scale = oracle.spspmm(iA, np.abs(vA), iB, np.abs(vB), m, k, n)
        assert np.all(np.abs(val.cpu().numpy() - ref_val) <= 1e-5 * scale)


def test_value_less_and_empty_operands():
    from paddle_sparse_amd import spspmm

    rng = np.random.default_rng(3)
    iA, vA = random_coo(200, 150, 2000, rng)
    iB, vB = random_coo(150, 100, 1500, rng)
    ref_idx, ref_val = oracle.spspmm(iA, vA, iB, None, 200, 150, 100)
    idx, val = spspmm(dev(iA), dev(vA), dev(iB), None, 200, 150, 100)
    assert np.array_equal(idx.cpu().numpy(), ref_idx) and np.array_equal(val.cpu().numpy(), ref_val)
    idx, val = spspmm(dev(iA), None, dev(iB), None, 200, 150, 100)
    assert np.array_equal(idx.cpu().numpy(), ref_idx) and val is None
    empty = torch.empty((2, 0), dtype=torch.int64, device="cuda")
    idx, val = spspmm(empty, torch.empty(0, device="cuda"), dev(iB), dev(vB), 200, 150, 100)
    assert idx.shape == (2, 0) and val.numel() == 0
    # A only points at empty rows of B: no products at all
    idx, val = spspmm(dev(np.array([[0], [5]])), dev(np.ones(1, np.float32)),
                      dev(np.array([[1], [2]])), dev(np.ones(1, np.float32)), 3, 8, 4)
    assert idx.shape == (2, 0) and val.numel() == 0


def test_coalesced_flag_and_tensor_matmul():
    from paddle_sparse_amd import SparseTensor, spspmm

    rng = np.random.default_rng(4)
    iA, vA = random_coo(300, 200, 4000, rng)
    iB, vB = random_coo(200, 250, 3000, rng)
    ref_idx, ref_val = oracle.spspmm(iA, vA, iB, vB, 300, 200, 250)
    # shuffled operands + coalesced=True
    pA, pB = rng.permutation(vA.size), rng.permutation(vB.size)
    idx, val = spspmm(dev(iA[:, pA]), dev(vA[pA]), dev(iB[:, pB]), dev(vB[pB]), 300, 200, 250, coalesced=True)
    assert np.array_equal(idx.cpu().numpy(), ref_idx) and np.array_equal(val.cpu().numpy(), ref_val)
    A = SparseTensor(row=dev(iA[0]), col=dev(iA[1]), value=dev(vA), sparse_sizes=(300, 200))
    B = SparseTensor(row=dev(iB[0]), col=dev(iB[1]), value=dev(vB), sparse_sizes=(200, 250))
    C = A @ B
    row, col, value = C.coo()
    assert C.sparse_sizes() == (300, 250)
    assert np.array_equal(torch.stack([row, col]).cpu().numpy(), ref_idx)
    assert np.array_equal(value.cpu().numpy(), ref_val)
    with pytest.raises(NotImplementedError):
        A.matmul(B, reduce="max")


def test_square_of_a_graph_matches_dense():
    """A @ A on a 2000-node graph against the dense fp64 product."""
    from paddle_sparse_amd import spspmm

    rng = np.random.default_rng(5)
    iA, vA = random_coo(2000, 2000, 30_000, rng)
    idx, val = spspmm(dev(iA), dev(vA), dev(iA), dev(vA), 2000, 2000, 2000)
    dense = np.zeros((2000, 2000))
    dense[iA[0], iA[1]] = vA
    ref = dense @ dense
    got = np.zeros_like(ref)
    got[idx[0].cpu().numpy(), idx[1].cpu().numpy()] = val.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_vs_frozen_scipy_products(tag, dtype):
    """tests/golden/third_party.npz: scipy.sparse CSR @ CSR on seeded matrices, structural product
    (entries whose terms cancel stay stored), float64 answers; both routes of the HIP path — the
    column walk for 4-byte values and the row-order walk for 8-byte ones — are held to them."""
    from pathlib import Path

    from paddle_sparse_amd import spspmm

    g = np.load(Path(__file__).resolve().parent / "golden" / "third_party.npz")
    m, k, n = (int(x) for x in g[f"spspmm_{tag}_shape"])
    idx, val = spspmm(dev(g[f"spspmm_{tag}_indexA"]), dev(g[f"spspmm_{tag}_valueA"], dtype),
                      dev(g[f"spspmm_{tag}_indexB"]), dev(g[f"spspmm_{tag}_valueB"], dtype), m, k, n)
    assert np.array_equal(idx.cpu().numpy(), g[f"spspmm_{tag}_index"])
    got, want, scale = val.cpu().numpy().astype(np.float64), g[f"spspmm_{tag}_value"], g[f"spspmm_{tag}_abs"]
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert np.all(np.abs(got - want) <= tol * scale + 1e-30)
    if tag == "c":  # integer-valued operands: exact in either width
        assert np.array_equal(got, want) and int((got == 0).sum()) == int(g["spspmm_c_cancelled"][0])
