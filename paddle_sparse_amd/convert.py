"""Host-side format conversions — paddle_sparse/convert.py:9-28.

No kernels here (the reference has none either): framework sparse tensors and
scipy matrices in and out of the (index, value) form.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse
import torch


def to_torch_sparse(index, value, m, n):
    """Counterpart of to_paddle_sparse (convert.py:9-10)."""
    return torch.sparse_coo_tensor(index.detach(), value, (m, n))


def from_torch_sparse(A):
    """Counterpart of from_paddle_sparse (convert.py:13-14)."""
    A = A.coalesce() if not A.is_coalesced() else A
    return A.indices().detach(), A.values()


def to_scipy(index, value, m, n):
    assert not index.is_cuda and not value.is_cuda
    (row, col), data = index.detach().numpy(), value.detach().numpy()
    return scipy.sparse.coo_matrix((data, (row, col)), (m, n))


def from_scipy(A, device=None):
    A = A.tocoo()
    row = torch.from_numpy(A.row.astype(np.int64))
    col = torch.from_numpy(A.col.astype(np.int64))
    value = torch.from_numpy(A.data)
    index = torch.stack([row, col], dim=0)
    if device is not None:
        index, value = index.to(device), value.to(device)
    return index, value
