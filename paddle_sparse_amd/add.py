"""add — paddle_sparse/add.py:12-100.

sparse + sparse is "concatenate and coalesce" in the reference (add.py:30-47).
Both operands are in (row, col) order already, so the two key streams are
MERGED (ops.merge_sorted) instead of sorted, then run lengths + segmented add
as in coalesce.py; tiny inputs take the one-workgroup coalesce instead.  sparse + dense vector broadcasts along rows or
columns through the HIP row gather (elementwise.py).
"""
from __future__ import annotations

import torch

from . import elementwise as ew
from .coalesce import _ONE_WORKGROUP_BELOW, _coalesce_sorted_stream, _coalesce_two_sorted
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
    if ra.numel() + rb.numel() > _ONE_WORKGROUP_BELOW:
        # both operands are in (row, col) order: merge, don't sort
        row, col, both = _coalesce_two_sorted(ra, ca, va, rb, cb, vb, shape[1], "sum")
    else:
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
