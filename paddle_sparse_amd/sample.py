"""sample / sample_adj / permute — paddle_sparse/sample.py:10-57,
paddle_sparse/permute.py:8-13.

sample_adj runs on the GPU here (the reference's op raises "No CUDA version
supported" for GPU tensors, csrc/sample.cpp:13-18): ops.sample_adj chains the
HIP kernels of csrc/sample.hip.  Its random picks come from a counter-based
stream selected by `seed`; by default a fresh seed is taken from torch's
global generator, so torch.manual_seed() makes runs repeatable the way the
reference's framework-wide seed does.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from .tensor import SparseTensor


def sample(src: SparseTensor, num_neighbors: int, subset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sample.py:10-27: [rows, num_neighbors] column ids drawn with replacement
    from each row's neighbours."""
    rowptr, col, _ = src.csr()
    rowcount = src.storage.rowcount()
    if subset is not None:
        rowcount, start = rowcount[subset], rowptr[subset]
    else:
        start = rowptr[:-1]
    rand = torch.rand((rowcount.shape[0], num_neighbors), device=col.device)
    rand = (rand * rowcount.to(rand.dtype).view(-1, 1)).to(torch.int64)
    # float rounding can land exactly on rowcount: keep the pick inside the row
    rand = torch.minimum(rand, (rowcount.view(-1, 1) - 1).clamp_(min=0))
    rand += start.view(-1, 1)
    if col.numel() > 0:
        rand.clamp_(max=col.numel() - 1)  # empty trailing rows would index past the end
    return col[rand]


def sample_adj(src: SparseTensor, subset: torch.Tensor, num_neighbors: int, replace: bool = False,
               seed: Optional[int] = None) -> Tuple[SparseTensor, torch.Tensor]:
    """sample.py:30-53: the sampled, relabelled sub-adjacency of `subset`
    ([len(subset), len(n_id)]) and the node ids n_id its columns refer to."""
    rowptr, col, value = src.csr()
    if seed is None:
        seed = int(torch.randint(0, 2**62, (1,)).item())
    out_rowptr, out_col, n_id, e_id = ops.sample_adj(
        rowptr, col, subset, num_neighbors, replace, seed=seed, num_cols=src.sparse_size(1))
    if value is not None:
        value = ops.gather_rows(value, e_id)
    out = SparseTensor(rowptr=out_rowptr, row=None, col=out_col, value=value,
                       sparse_sizes=(subset.shape[0], n_id.shape[0]), is_sorted=True, trust_data=True)
    return out, n_id


def permute(src: SparseTensor, perm: torch.Tensor) -> SparseTensor:
    """permute.py:8-10: A[perm][:, perm] of a square matrix."""
    assert src.is_quadratic()
    return src.index_select(0, perm).index_select(1, perm)


SparseTensor.sample = sample
SparseTensor.sample_adj = sample_adj
SparseTensor.permute = lambda self, perm: permute(self, perm)
