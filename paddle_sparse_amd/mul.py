"""mul — paddle_sparse/mul.py:12-128.

sparse * sparse keeps the entries present in BOTH (coalesced) operands: the
reference concatenates, argsorts the keys and multiplies neighbours with equal
keys (mul.py:57-73).  Both key streams are sorted already, so a stable merge
(ops.merge_sorted) replaces the sort; with both inputs coalesced a key occurs
at most twice, and the merge puts A's entry first.
"""
from __future__ import annotations

import torch

from . import elementwise as ew
from . import ops
from .tensor import SparseTensor

_mul_dense = ew.dense_variant("mul", inplace=False)
mul_ = ew.dense_variant("mul", inplace=True)
mul_nnz = ew.nnz_variant("mul", inplace=False)
mul_nnz_ = ew.nnz_variant("mul", inplace=True)


def _mul_sparse(a: SparseTensor, b: SparseTensor) -> SparseTensor:
    for name, t in (("src", a), ("other", b)):
        if not t.is_coalesced():
            raise ValueError(f"The `{name}` tensor is not coalesced")
    (ra, ca, va), (rb, cb, vb) = a.coo(), b.coo()
    if va is None or vb is None:
        raise ValueError("Both sparse tensors must contain values")
    M, N = max(a.size(0), b.size(0)), max(a.size(1), b.size(1))
    keys_a, _ = ops.make_keys(ra, ca, N)
    keys_b, _ = ops.make_keys(rb, cb, N)
    rides = (va.dim() == 1 and vb.dim() == 1 and va.element_size() == 4 and va.dtype == vb.dtype
             and not ops.needs_grad(va) and not ops.needs_grad(vb))
    keys, source, value = ops.merge_sorted(keys_a, keys_b, va.contiguous() if rides else None,
                                           vb.contiguous() if rides else None, want_source=not rides)
    if not rides:
        value = ops.gather_rows(torch.cat([va, vb], dim=0), source)
    first = (keys[1:] == keys[:-1]).nonzero().view(-1)  # A's entry of every key both operands hold
    row, col = ops.split_keys(keys[first], N)
    return SparseTensor(row=row, col=col, value=value[first] * value[first + 1], sparse_sizes=(M, N),
                        is_sorted=True, trust_data=True)


def mul(src: SparseTensor, other):
    if isinstance(other, torch.Tensor):
        return _mul_dense(src, other)
    assert isinstance(other, SparseTensor)
    return _mul_sparse(src, other)


ew.install("mul", mul, mul_, mul_nnz, mul_nnz_)
