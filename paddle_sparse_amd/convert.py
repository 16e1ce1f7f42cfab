"""Host-side format conversions and the functional `eye`
(paddle_sparse/convert.py:9-28, paddle_sparse/eye.py:6-24).

Nothing here touches the HIP core: these are the doors between the (index,
value) COO form of the functional API and the host frameworks' own sparse
types.  The framework on this side of the boundary is torch, so the
`*_paddle_sparse` pair of the reference is the `*_torch_sparse` pair here.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import scipy.sparse
import torch

IndexValue = Tuple[torch.Tensor, torch.Tensor]


def _check_coo(index: torch.Tensor, value: torch.Tensor) -> None:
    if index.dim() != 2 or index.shape[0] != 2:
        raise ValueError(f"index must be [2, nnz] (got {tuple(index.shape)})")
    if value.shape[0] != index.shape[1]:
        raise ValueError("value needs one entry per index column")


def to_torch_sparse(index: torch.Tensor, value: torch.Tensor, m: int, n: int) -> torch.Tensor:
    """(index, value) of an m x n matrix -> torch sparse COO tensor (the
    reference's to_paddle_sparse)."""
    _check_coo(index, value)
    return torch.sparse_coo_tensor(index.detach(), value, size=(m, n) + tuple(value.shape[1:]))


def from_torch_sparse(A: torch.Tensor) -> IndexValue:
    """torch sparse COO tensor -> (index, value), duplicates merged (the
    reference's from_paddle_sparse)."""
    if not A.is_coalesced():
        A = A.coalesce()
    return A.indices().detach(), A.values()


# The reference names these after its host framework's sparse COO type
# (convert.py:9-14); callers written against it keep working.
to_paddle_sparse = to_torch_sparse
from_paddle_sparse = from_torch_sparse


def to_scipy(index: torch.Tensor, value: torch.Tensor, m: int, n: int) -> scipy.sparse.coo_matrix:
    """Host tensors only, as in the reference (convert.py:18): move GPU data to
    the CPU first."""
    _check_coo(index, value)
    if index.is_cuda or value.is_cuda:
        raise AssertionError("to_scipy needs CPU tensors (call .cpu() first)")
    coords = index.detach().numpy()
    return scipy.sparse.coo_matrix((value.detach().numpy(), (coords[0], coords[1])), shape=(m, n))


def from_scipy(A, device: Optional[torch.device] = None) -> IndexValue:
    """Any scipy sparse matrix -> (index int64[2, nnz], value); `device` moves
    the pair to HBM in the same call."""
    coo = A.tocoo()
    index = torch.from_numpy(np.stack([coo.row, coo.col]).astype(np.int64, copy=False))
    value = torch.from_numpy(np.ascontiguousarray(coo.data))
    return (index, value) if device is None else (index.to(device), value.to(device))


def eye(m: int, dtype: Optional[torch.dtype] = None, device=None) -> IndexValue:
    """Identity of size m as (index, value): ones on the diagonal (eye.py:6-24)."""
    diag = torch.arange(m, dtype=torch.int64, device=device)
    return diag.repeat(2, 1), torch.ones(m, dtype=dtype, device=device)
