"""mul — paddle_sparse/mul.py:12-128.

sparse * sparse keeps the entries present in BOTH (coalesced) operands: the
reference concatenates, argsorts the keys and multiplies neighbours with equal
keys (mul.py:57-73).  Same steps here on the HIP radix sort; with both inputs
coalesced a key occurs at most twice, and the stable sort puts A's entry first.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .add import _broadcast_operand
from .tensor import SparseTensor


def mul(src: SparseTensor, other):
    if isinstance(other, torch.Tensor):
        picked = _broadcast_operand(src, other)
        value = src.storage.value()
        value = picked.to(value.dtype) * value if value is not None else picked
        return src.set_value(value, layout="coo")

    assert isinstance(other, SparseTensor)
    if not src.is_coalesced():
        raise ValueError("The `src` tensor is not coalesced")
    if not other.is_coalesced():
        raise ValueError("The `other` tensor is not coalesced")
    row_a, col_a, value_a = src.coo()
    row_b, col_b, value_b = other.coo()
    if value_a is None or value_b is None:
        raise ValueError("Both sparse tensors must contain values")
    M = max(src.size(0), other.size(0))
    N = max(src.size(1), other.size(1))
    value = torch.cat([value_a, value_b], dim=0)
    keys, _ = ops.make_keys(torch.cat([row_a, row_b]), torch.cat([col_a, col_b]), N)
    keys, perm = ops.index_sort(keys, M * N, with_sorted_inputs=True)
    value = ops.gather_rows(value, perm)
    hit = (keys[1:] == keys[:-1]).nonzero().view(-1)  # position of A's entry of each common key
    common = keys[hit]
    row = torch.div(common, N, rounding_mode="floor")
    return SparseTensor(row=row, col=common - row * N, value=value[hit] * value[hit + 1],
                        sparse_sizes=(M, N), is_sorted=True, trust_data=True)


def mul_(src: SparseTensor, other: torch.Tensor) -> SparseTensor:
    picked = _broadcast_operand(src, other)
    value = src.storage.value()
    value = value.mul_(picked.to(value.dtype)) if value is not None else picked
    return src.set_value_(value, layout="coo")


def mul_nnz(src: SparseTensor, other: torch.Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = value * other.to(value.dtype) if value is not None else other
    return src.set_value(value, layout=layout)


def mul_nnz_(src: SparseTensor, other: torch.Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = value.mul_(other.to(value.dtype)) if value is not None else other
    return src.set_value_(value, layout=layout)


SparseTensor.mul = lambda self, other: mul(self, other)
SparseTensor.mul_ = lambda self, other: mul_(self, other)
SparseTensor.mul_nnz = lambda self, other, layout=None: mul_nnz(self, other, layout)
SparseTensor.mul_nnz_ = lambda self, other, layout=None: mul_nnz_(self, other, layout)
SparseTensor.__mul__ = SparseTensor.mul
SparseTensor.__rmul__ = SparseTensor.mul
SparseTensor.__imul__ = SparseTensor.mul_
