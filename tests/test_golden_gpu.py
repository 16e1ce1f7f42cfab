"""GPU: the HIP path against the frozen third-party answers of
tests/golden/third_party.npz (torch-CPU torch.sparse.mm forward + autograd,
numpy ufunc.at, torch.segment_reduce — see tests/golden/make_golden.py) for the
rows the reference holds no fixture for: SpMM forward / backward on both kernel
families, reduce over dim 0 / 1, coalesce with every reduction.
Tolerance for fp32 sums: 1e-5 of the sum of the absolute terms (north star)."""
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def golden():
    return np.load(ROOT / "tests" / "golden" / "third_party.npz")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("algo", ["row_waves", "edge_ranges"])
def test_spmm_forward_backward_vs_golden(golden, tag, reduce, algo):
    from paddle_sparse_amd import SparseTensor

    g = golden
    rowptr, col, val, B, G = (g[f"spmm_{tag}_{k}"] for k in ("rowptr", "col", "val", "B", "G"))
    M, N = rowptr.size - 1, B.shape[0]
    row = np.repeat(np.arange(M, dtype=np.int64), np.diff(rowptr))
    v = dev(val).requires_grad_()
    Bt = dev(B).requires_grad_()
    a = SparseTensor(row=dev(row), col=dev(col), value=v, sparse_sizes=(M, N), is_sorted=True)
    a.storage._spmm_algo_memo = algo  # pin the forward kernel family for this test
    out = a.matmul(Bt, reduce)
    out.backward(dev(G))
    S = oracle.spmm_abs_sum(rowptr, col, val, B)
    assert np.all(np.abs(out.detach().cpu().numpy() - g[f"spmm_{tag}_{reduce}_out"]) <= 1e-5 * S + 1e-30)
    if reduce in ("sum", "mean"):
        sm = oracle.spmm_mat_bw(reduce, row, rowptr, col, np.abs(val), np.abs(G), N)
        sv = oracle.spmm_value_bw(reduce, row, rowptr, col, np.abs(B), np.abs(G))
    else:
        arg = oracle.spmm(reduce, rowptr, col, val, B)[1]
        sv, sm = oracle.spmm_minmax_bw(col, np.abs(val), np.abs(B), np.abs(G), arg)
    assert np.all(np.abs(Bt.grad.cpu().numpy() - g[f"spmm_{tag}_{reduce}_gmat"]) <= 1e-5 * sm + 1e-30)
    assert np.all(np.abs(v.grad.cpu().numpy() - g[f"spmm_{tag}_{reduce}_gval"]) <= 1e-5 * sv + 1e-30)


def test_reduce_dim0_dim1_vs_golden(golden):
    from paddle_sparse_amd import SparseTensor

    g = golden
    M, N = (int(x) for x in g["red_shape"])
    for warm_csc in (False, True):  # dim 0: scatter (cold) and segment (CSC caches present) paths
        t = SparseTensor(row=dev(g["red_row"]), col=dev(g["red_col"]), value=dev(g["red_val"]), sparse_sizes=(M, N))
        if warm_csc:
            t.storage.csr2csc()
        for dim in (0, 1):
            for reduce in ("sum", "mean", "min", "max"):
                got = getattr(t, reduce)(dim).cpu().numpy()
                ref = g[f"red_dim{dim}_{reduce}"]
                if reduce in ("min", "max"):
                    assert np.array_equal(got, ref), (dim, reduce)
                else:
                    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)


def test_coalesce_vs_golden(golden):
    import paddle_sparse_amd as ps

    g = golden
    m, n = (int(x) for x in g["co_shape"])
    index = dev(np.stack([g["co_row"], g["co_col"]]))
    for op in ("add", "mean", "min", "max"):
        gi, gv = ps.coalesce(index, dev(g["co_val"]), m, n, op)
        assert np.array_equal(gi.cpu().numpy(), g["co_index"])
        if op in ("min", "max"):
            assert np.array_equal(gv.cpu().numpy(), g[f"co_{op}"])
        else:
            np.testing.assert_allclose(gv.cpu().numpy(), g[f"co_{op}"], rtol=1e-5, atol=1e-6)
