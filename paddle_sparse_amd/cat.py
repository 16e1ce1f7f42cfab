"""cat(tensors, dim) — paddle_sparse/cat.py:12-276.

dim = 0 stacks rows, dim = 1 stacks columns, dim = (0, 1) builds the block
diagonal, dim >= 2 concatenates the dense value dimensions.  A cache of the
result is kept exactly when every operand carries it (cat.py:82-100,
143-153, 224-254); index arithmetic only, no kernel of its own.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Union

import torch

from .storage import SparseStorage
from .tensor import SparseTensor


def _gather(tensors: Sequence[SparseTensor], field: str,
            shift: Optional[Callable[[torch.Tensor, int, int, int, bool], torch.Tensor]] = None
            ) -> Optional[torch.Tensor]:
    """Concatenation of storage.<field> over all operands, or None unless every
    operand has it.  `shift(x, row_offset, col_offset, nnz_offset, first)` maps
    one operand's array into the result's numbering."""
    parts: List[torch.Tensor] = []
    rows = cols = nnz = 0
    for t in tensors:
        x = getattr(t.storage, field)
        if x is None:
            return None
        parts.append(x if shift is None else shift(x, rows, cols, nnz, len(parts) == 0))
        rows, cols, nnz = rows + t.sparse_size(0), cols + t.sparse_size(1), nnz + t.nnz()
    return torch.cat(parts, dim=0)


def _ptr_shift(x, _r, _c, nnz, first):
    # pointer arrays chain: drop the leading 0 of every operand but the first
    return x if first else x[1:] + nnz


def _row_or_rowptr(tensors, row_shift):
    row = _gather(tensors, "_row", row_shift)
    rowptr = _gather(tensors, "_rowptr", _ptr_shift)
    if row is None and rowptr is None:  # mixed layouts: expand the missing rows
        for t in tensors:
            t.storage.row()
        row = _gather(tensors, "_row", row_shift)
    return row, rowptr


def cat_first(tensors: Sequence[SparseTensor]) -> SparseTensor:
    row, rowptr = _row_or_rowptr(tensors, lambda x, r, c, n, f: x + r)
    storage = SparseStorage(
        row=row, rowptr=rowptr,
        col=_gather(tensors, "_col"),
        value=_gather(tensors, "_value"),
        sparse_sizes=(sum(t.sparse_size(0) for t in tensors), max(t.sparse_size(1) for t in tensors)),
        rowcount=_gather(tensors, "_rowcount"),
        is_sorted=True, trust_data=True)
    return tensors[0].from_storage(storage)


def cat_second(tensors: Sequence[SparseTensor]) -> SparseTensor:
    for t in tensors:
        t.storage.row()
    storage = SparseStorage(
        row=_gather(tensors, "_row"),
        col=_gather(tensors, "_col", lambda x, r, c, n, f: x + c),
        value=_gather(tensors, "_value"),
        sparse_sizes=(max(t.sparse_size(0) for t in tensors), sum(t.sparse_size(1) for t in tensors)),
        colptr=_gather(tensors, "_colptr", _ptr_shift),
        colcount=_gather(tensors, "_colcount"),
        is_sorted=False, trust_data=True)  # rows interleave: the constructor sorts (cat.py:163)
    return tensors[0].from_storage(storage)


def cat_diag(tensors: Sequence[SparseTensor]) -> SparseTensor:
    row, rowptr = _row_or_rowptr(tensors, lambda x, r, c, n, f: x + r)
    perm_shift = lambda x, r, c, n, f: x + n  # noqa: E731
    storage = SparseStorage(
        row=row, rowptr=rowptr,
        col=_gather(tensors, "_col", lambda x, r, c, n, f: x + c),
        value=_gather(tensors, "_value"),
        sparse_sizes=(sum(t.sparse_size(0) for t in tensors), sum(t.sparse_size(1) for t in tensors)),
        rowcount=_gather(tensors, "_rowcount"),
        colptr=_gather(tensors, "_colptr", _ptr_shift),
        colcount=_gather(tensors, "_colcount"),
        csr2csc=_gather(tensors, "_csr2csc", perm_shift),
        csc2csr=_gather(tensors, "_csc2csr", perm_shift),
        is_sorted=True, trust_data=True)
    return tensors[0].from_storage(storage)


def cat(tensors: Sequence[SparseTensor], dim: Union[int, Sequence[int]]) -> SparseTensor:
    assert len(tensors) > 0
    if not isinstance(dim, int):
        assert isinstance(dim, (tuple, list)) and sorted(dim) == [0, 1]
        return cat_diag(tensors)
    ndim = tensors[0].dim()
    if dim < 0:
        dim += ndim
    if dim == 0:
        return cat_first(tensors)
    if dim == 1:
        return cat_second(tensors)
    if 1 < dim < ndim:
        values = [t.storage.value() for t in tensors]
        assert all(v is not None for v in values)
        return tensors[0].set_value(torch.cat(values, dim=dim - 1), layout="coo")
    raise IndexError(f"Dimension out of range: Expected to be in range of [{-ndim}, {ndim - 1}], "
                     f"but got {dim}.")
