"""GPU: the usage section of README.md runs as written (same calls, small data)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_readme_usage():
    import paddle_sparse_amd as ps
    from paddle_sparse_amd import SparseTensor

    index = torch.tensor([[1, 0, 1, 0, 2, 1], [0, 1, 1, 1, 0, 0]], device="cuda")
    value = torch.tensor([[1., 2.], [2., 3.], [3., 4.], [4., 5.], [5., 6.], [6., 7.]], device="cuda")
    index, value = ps.coalesce(index, value, m=3, n=2)
    assert index.tolist() == [[0, 1, 1, 2], [1, 0, 1, 0]] and value.tolist() == [[6, 8], [7, 9], [3, 4], [5, 6]]
    index, value = ps.transpose(index, value, 3, 2)
    assert index.tolist() == [[0, 0, 1, 1], [1, 2, 0, 1]] and value.tolist() == [[7, 9], [5, 6], [6, 8], [3, 4]]

    g = torch.Generator(device="cuda").manual_seed(0)
    N, F = 500, 16
    key = torch.unique(torch.randint(0, N * N, (4000,), generator=g, device="cuda"))
    row, col = torch.div(key, N, rounding_mode="floor"), key % N
    w = torch.rand(row.numel(), generator=g, device="cuda").requires_grad_()
    x = torch.randn(N, F, generator=g, device="cuda", requires_grad=True)
    adj = SparseTensor(row=row, col=col, value=w, sparse_sizes=(N, N))
    out = adj @ x
    out.sum().backward()
    dense = torch.zeros(N, N, device="cuda").index_put((row, col), w.detach())
    assert torch.allclose(out, dense @ x.detach(), atol=1e-4)
    assert torch.allclose(x.grad, dense.t() @ torch.ones(N, F, device="cuda"), atol=1e-4) and w.grad.shape == w.shape
    for reduce in ("mean", "min", "max"):
        assert adj.matmul(x, reduce=reduce).shape == (N, F)
    adj_t = adj.t()
    assert torch.allclose(adj_t.to_dense(), dense.t())
    assert torch.allclose(adj.sum(dim=1), dense.sum(1), atol=1e-4)
    batch = torch.arange(0, 64, device="cuda")
    sub, n_id = adj.sample_adj(batch, num_neighbors=5)
    assert sub.sparse_sizes() == (64, n_id.numel()) and torch.equal(n_id[:64], batch)
    two_hop = adj.detach() @ adj.detach()
    assert torch.allclose(two_hop.to_dense(), dense @ dense, atol=1e-3)
