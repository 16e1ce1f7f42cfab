"""Element-wise add / mul between a SparseTensor and a dense vector or a
per-entry tensor (paddle_sparse/add.py:12-28,50-100, mul.py:12-33,76-128).

The eight public variants (add, add_, add_nnz, add_nnz_, mul, mul_, mul_nnz,
mul_nnz_) differ only in the arithmetic, in whether the operand is first
spread over the stored entries, and in whether the value array is updated in
place; they are produced from one table here instead of being spelled out.
A value-less matrix counts as all ones (add.py:20-23, mul.py:22-25).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from . import ops
from .tensor import SparseTensor

# name -> (out-of-place, in-place, result when the matrix has no values)
_ARITH: Dict[str, Tuple[Callable, Callable, Callable]] = {
    "add": (lambda v, x: v + x, lambda v, x: v.add_(x), lambda x: x + 1),
    "mul": (lambda v, x: v * x, lambda v, x: v.mul_(x), lambda x: x),
}


def spread(src: SparseTensor, other: torch.Tensor) -> torch.Tensor:
    """One operand entry per stored entry: other[row] for an (M, 1) column
    vector, other[col] for a (1, N) row vector — a HIP row gather in place of
    the reference's gather_csr / fancy index (add.py:15-18, mul.py:15-20)."""
    m, n = src.size(0), src.size(1)
    if other.dim() == 2 and tuple(other.shape) == (m, 1):
        return ops.gather_rows(other.reshape(m).contiguous(), src.storage.row())
    if other.dim() == 2 and tuple(other.shape) == (1, n):
        return ops.gather_rows(other.reshape(n).contiguous(), src.storage.col())
    raise ValueError(f"Size mismatch: Expected size ({m}, 1, ...) or (1, {n}, ...), "
                     f"but got size {tuple(other.shape)}.")


def _apply(kind: str, src: SparseTensor, operand: torch.Tensor, inplace: bool, layout: Optional[str]):
    fresh, update, no_value = _ARITH[kind]
    value = src.storage.value()
    if value is None:
        new = no_value(operand)
    else:
        new = (update if inplace else fresh)(value, operand.to(value.dtype))
    return (src.set_value_ if inplace else src.set_value)(new, layout=layout)


def dense_variant(kind: str, inplace: bool):
    """add/mul with a broadcast dense operand (layout of the result: coo)."""
    def fn(src: SparseTensor, other: torch.Tensor) -> SparseTensor:
        return _apply(kind, src, spread(src, other), inplace, "coo")
    fn.__name__ = kind + ("_" if inplace else "")
    return fn


def nnz_variant(kind: str, inplace: bool):
    """add_nnz/mul_nnz: `other` already holds one entry per stored entry."""
    def fn(src: SparseTensor, other: torch.Tensor, layout: Optional[str] = None) -> SparseTensor:
        return _apply(kind, src, other, inplace, layout)
    fn.__name__ = f"{kind}_nnz" + ("_" if inplace else "")
    return fn


def install(kind: str, binary: Callable, inplace_binary: Callable, nnz: Callable, nnz_inplace: Callable) -> None:
    """Method and operator forms on SparseTensor (add.py:90-100, mul.py:118-128)."""
    setattr(SparseTensor, kind, lambda self, other: binary(self, other))
    setattr(SparseTensor, kind + "_", lambda self, other: inplace_binary(self, other))
    setattr(SparseTensor, kind + "_nnz", lambda self, other, layout=None: nnz(self, other, layout))
    setattr(SparseTensor, kind + "_nnz_", lambda self, other, layout=None: nnz_inplace(self, other, layout))
    setattr(SparseTensor, f"__{kind}__", getattr(SparseTensor, kind))
    setattr(SparseTensor, f"__r{kind}__", getattr(SparseTensor, kind))
    setattr(SparseTensor, f"__i{kind}__", getattr(SparseTensor, kind + "_"))
