"""add — paddle_sparse/add.py:12-100.

sparse + sparse is "concatenate and coalesce" in the reference (add.py:30-47);
here the concatenation feeds the fused sort / run-length / segmented-add chain
of coalesce.py directly.  sparse + dense vector broadcasts along rows or
columns through the HIP row gather (elementwise.py).
"""
from __future__ import annotations

import torch

from . import elementwise as ew
from .coalesce import _coalesce_sorted_stream
from .tensor import SparseTensor

_add_dense = ew.dense_variant("add", inplace=False)
add_ = ew.dense_variant("add", inplace=True)
add_nnz = ew.nnz_variant("add", inplace=False)
add_nnz_ = ew.nnz_variant("add", inplace=True)


def _add_sparse(a: SparseTensor, b: SparseTensor) -> SparseTensor:
    """Union of the stored entries, values of shared entries added.  The result
    has values only when both operands do (add.py:37-39)."""
    shape = (max(a.size(0), b.size(0)), max(a.size(1), b.size(1)))
    (ra, ca, va), (rb, cb, vb) = a.coo(), b.coo()
    both = None if va is None or vb is None else torch.cat([va, vb], dim=0)
    row, col, both = _coalesce_sorted_stream(torch.cat([ra, rb]), torch.cat([ca, cb]), both,
                                             shape[0], shape[1], "sum")
    return SparseTensor(row=row, col=col, value=both, sparse_sizes=shape, is_sorted=True, trust_data=True)


def add(src: SparseTensor, other):
    if isinstance(other, SparseTensor):
        return _add_sparse(src, other)
    if isinstance(other, torch.Tensor):
        return _add_dense(src, other)
    raise NotImplementedError


ew.install("add", add, add_, add_nnz, add_nnz_)
