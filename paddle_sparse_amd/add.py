"""add — paddle_sparse/add.py:12-100.

sparse + sparse is "concatenate and coalesce" in the reference (add.py:30-47);
here the concatenation feeds the fused sort / run-length / segmented-add chain
of coalesce.py directly.  sparse + dense vector broadcasts along rows or
columns through the HIP row gather.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .coalesce import _coalesce_sorted_stream
from .tensor import SparseTensor


def _broadcast_operand(src: SparseTensor, other: torch.Tensor) -> torch.Tensor:
    """other[row] for an (M, 1) operand, other[col] for a (1, N) one
    (gather_csr / index at add.py:15-18, mul.py:15-20)."""
    if other.dim() == 2 and other.shape[0] == src.size(0) and other.shape[1] == 1:
        return ops.gather_rows(other.squeeze(1).contiguous(), src.storage.row())
    if other.dim() == 2 and other.shape[0] == 1 and other.shape[1] == src.size(1):
        return ops.gather_rows(other.squeeze(0).contiguous(), src.storage.col())
    raise ValueError(f"Size mismatch: Expected size ({src.size(0)}, 1, ...) or "
                     f"(1, {src.size(1)}, ...), but got size {tuple(other.shape)}.")


def add(src: SparseTensor, other):
    if isinstance(other, torch.Tensor):
        picked = _broadcast_operand(src, other)
        value = src.storage.value()
        value = picked.to(value.dtype) + value if value is not None else picked + 1
        return src.set_value(value, layout="coo")
    if isinstance(other, SparseTensor):
        row_a, col_a, value_a = src.coo()
        row_b, col_b, value_b = other.coo()
        M = max(src.size(0), other.size(0))
        N = max(src.size(1), other.size(1))
        value: Optional[torch.Tensor] = None
        if value_a is not None and value_b is not None:
            value = torch.cat([value_a, value_b], dim=0)
        row, col, value = _coalesce_sorted_stream(torch.cat([row_a, row_b]), torch.cat([col_a, col_b]),
                                                  value, M, N, "sum")
        return SparseTensor(row=row, col=col, value=value, sparse_sizes=(M, N),
                            is_sorted=True, trust_data=True)
    raise NotImplementedError


def add_(src: SparseTensor, other: torch.Tensor) -> SparseTensor:
    picked = _broadcast_operand(src, other)
    value = src.storage.value()
    value = value.add_(picked.to(value.dtype)) if value is not None else picked + 1
    return src.set_value_(value, layout="coo")


def add_nnz(src: SparseTensor, other: torch.Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = value + other.to(value.dtype) if value is not None else other + 1
    return src.set_value(value, layout=layout)


def add_nnz_(src: SparseTensor, other: torch.Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = value.add_(other.to(value.dtype)) if value is not None else other + 1
    return src.set_value_(value, layout=layout)


SparseTensor.add = lambda self, other: add(self, other)
SparseTensor.add_ = lambda self, other: add_(self, other)
SparseTensor.add_nnz = lambda self, other, layout=None: add_nnz(self, other, layout)
SparseTensor.add_nnz_ = lambda self, other, layout=None: add_nnz_(self, other, layout)
SparseTensor.__add__ = SparseTensor.add
SparseTensor.__radd__ = SparseTensor.add
SparseTensor.__iadd__ = SparseTensor.add_
