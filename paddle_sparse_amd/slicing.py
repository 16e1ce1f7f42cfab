"""narrow / select / index_select / masked_select and `SparseTensor[...]`
(the mini-batch gathers either side of SpMM).

Reference: paddle_sparse/narrow.py:11-100, select.py:7-11,
index_select.py:13-125, masked_select.py:12-119, tensor.py:704-757.  The
ragged row/column gathers (`repeat_interleave` + `gather_csr` there) run on the
HIP index kernels: count2ptr (new pointer), ptr2ind (new row of every kept
entry), gather_rows (col/value), index_sort (re-sort after a column gather).
Masks and O(#selected) arithmetic stay framework glue, as in the reference.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

from . import ops
from .storage import SparseStorage, get_layout
from .tensor import SparseTensor


def _ragged_take(old_ptr: torch.Tensor, counts: torch.Tensor, idx: torch.Tensor):
    """Positions of the entries of segments idx[0], idx[1], ... (in that
    order) of a pointer array: returns (new_ptr, new_segment_of_entry, perm)."""
    picked = ops.gather_rows(counts, idx)
    new_ptr = ops.count2ptr(picked)
    total = int(new_ptr[-1])
    seg = ops.ptr2ind(new_ptr, total)
    shift = ops.gather_rows(old_ptr, idx) - new_ptr[:-1]
    perm = torch.arange(total, dtype=torch.int64, device=idx.device) + ops.gather_rows(shift, seg)
    return picked, new_ptr, seg, perm


def index_select(src: SparseTensor, dim: int, idx: torch.Tensor) -> SparseTensor:
    dim = src.dim() + dim if dim < 0 else dim
    assert idx.dim() == 1
    idx = idx.to(torch.int64).contiguous()
    if dim == 0:  # index_select.py:17-49
        old_rowptr, col, value = src.csr()
        rowcount, rowptr, row, perm = _ragged_take(old_rowptr, src.storage.rowcount(), idx)
        storage = SparseStorage(row=row, rowptr=rowptr, col=ops.gather_rows(col, perm),
                                value=None if value is None else ops.gather_rows(value, perm),
                                sparse_sizes=(idx.numel(), src.sparse_size(1)), rowcount=rowcount,
                                is_sorted=True, trust_data=True)
        return src.from_storage(storage)
    if dim == 1:  # index_select.py:51-88
        old_colptr, row, value = src.csc()
        colcount, colptr, col, perm = _ragged_take(old_colptr, src.storage.colcount(), idx)
        row = ops.gather_rows(row, perm)
        keys, _ = ops.make_keys(row, col, idx.numel())
        _, csc2csr = ops.index_sort(keys, src.sparse_size(0) * max(idx.numel(), 1), check=True)
        if value is not None:
            value = ops.gather_rows(ops.gather_rows(value, perm), csc2csr)
        storage = SparseStorage(row=ops.gather_rows(row, csc2csr), col=ops.gather_rows(col, csc2csr),
                                value=value, sparse_sizes=(src.sparse_size(0), idx.numel()),
                                colptr=colptr, colcount=colcount, csc2csr=csc2csr,
                                is_sorted=True, trust_data=True)
        return src.from_storage(storage)
    value = src.storage.value()
    if value is None:
        raise ValueError
    return src.set_value(value.index_select(dim - 1, idx), layout="coo")


def index_select_nnz(src: SparseTensor, idx: torch.Tensor, layout: Optional[str] = None) -> SparseTensor:
    assert idx.dim() == 1
    idx = idx.to(torch.int64).contiguous()
    if get_layout(layout) == "csc":
        idx = ops.gather_rows(src.storage.csc2csr(), idx)
    row, col, value = src.coo()
    return SparseTensor(row=ops.gather_rows(row, idx), col=ops.gather_rows(col, idx),
                        value=None if value is None else ops.gather_rows(value, idx),
                        sparse_sizes=src.sparse_sizes(), is_sorted=True)


def masked_select(src: SparseTensor, dim: int, mask: torch.Tensor) -> SparseTensor:
    """masked_select.py:12-90: keeping the rows/columns where mask is set is
    an index_select with the (ascending) positions of the set bits."""
    dim = src.dim() + dim if dim < 0 else dim
    assert mask.dim() == 1 and mask.dtype == torch.bool
    if dim <= 1:
        return index_select(src, dim, mask.nonzero().view(-1))
    value = src.storage.value()
    if value is None:
        raise ValueError
    idx = mask.nonzero().view(-1)
    return src.set_value(value.index_select(dim - 1, idx), layout="coo")


def masked_select_nnz(src: SparseTensor, mask: torch.Tensor, layout: Optional[str] = None) -> SparseTensor:
    assert mask.dim() == 1
    if get_layout(layout) == "csc":
        mask = ops.gather_rows(mask.to(torch.uint8), src.storage.csc2csr()).to(torch.bool)
    row, col, value = src.coo()
    return SparseTensor(row=row[mask], col=col[mask], value=None if value is None else value[mask],
                        sparse_sizes=src.sparse_sizes(), is_sorted=True)


def narrow(src: SparseTensor, dim: int, start: int, length: int) -> SparseTensor:
    if dim < 0:
        dim = src.dim() + dim
    if start < 0:
        start = src.size(dim) + start
    st = src.storage
    if dim == 0:  # narrow.py:18-53: a contiguous slice of every CSR array
        rowptr, col, value = src.csr()
        rowptr = rowptr[start:start + length + 1]
        e0 = int(rowptr[0])
        rowptr = rowptr - e0
        e1 = e0 + int(rowptr[-1])
        row = st._row[e0:e1] - start if st._row is not None else None
        rowcount = st._rowcount[start:start + length] if st._rowcount is not None else None
        storage = SparseStorage(row=row, rowptr=rowptr, col=col[e0:e1],
                                value=None if value is None else value[e0:e1],
                                sparse_sizes=(length, src.sparse_size(1)), rowcount=rowcount,
                                is_sorted=True, trust_data=True)
        return src.from_storage(storage)
    if dim == 1:  # narrow.py:55-91
        row, col, value = src.coo()
        mask = (col >= start) & (col < start + length)
        colptr = st._colptr[start:start + length + 1] if st._colptr is not None else None
        if colptr is not None:
            colptr = colptr - colptr[0]
        colcount = st._colcount[start:start + length] if st._colcount is not None else None
        storage = SparseStorage(row=row[mask], col=col[mask] - start,
                                value=None if value is None else value[mask],
                                sparse_sizes=(src.sparse_size(0), length), colptr=colptr,
                                colcount=colcount, is_sorted=True, trust_data=True)
        return src.from_storage(storage)
    value = st.value()
    if value is None:
        raise ValueError
    return src.set_value(value.narrow(dim - 1, start, length), layout="coo")


def __narrow_diag__(src: SparseTensor, start, length) -> SparseTensor:
    """narrow.py:103-168 — one block of a block-diagonal matrix, i.e. the
    inverse of cat(..., dim=(0, 1)); only meaningful on such matrices (the
    block's entries are then one contiguous run of every array, in CSR and in
    CSC order alike, so each cache is a shifted slice)."""
    (r0, c0), (nr, nc) = start, length
    st = src.storage
    rowptr = st.rowptr()[r0:r0 + nr + 1]
    e0 = int(rowptr[0])
    rowptr = rowptr - e0
    e1 = e0 + int(rowptr[-1])

    def run(x, shift):  # the block's run of an nnz-sized array, renumbered
        return None if x is None else x[e0:e1] - shift

    colptr = st._colptr
    if colptr is not None:
        colptr = colptr[c0:c0 + nc + 1]
        colptr = colptr - colptr[0]
    storage = SparseStorage(
        row=run(st._row, r0), rowptr=rowptr, col=run(st._col, c0),
        value=None if st._value is None else st._value[e0:e1],
        sparse_sizes=(nr, nc),
        rowcount=None if st._rowcount is None else st._rowcount[r0:r0 + nr],
        colptr=colptr,
        colcount=None if st._colcount is None else st._colcount[c0:c0 + nc],
        csr2csc=run(st._csr2csc, e0), csc2csr=run(st._csc2csr, e0),
        is_sorted=True, trust_data=True)
    return src.from_storage(storage)


SparseTensor.__narrow_diag__ = lambda self, start, length: __narrow_diag__(self, start, length)


def select(src: SparseTensor, dim: int, idx: int) -> SparseTensor:
    return narrow(src, dim, start=idx, length=1)


def _getitem(self: SparseTensor, index: Any) -> SparseTensor:
    """tensor.py:704-757."""
    index = list(index) if isinstance(index, tuple) else [index]
    if len([i for i in index if not isinstance(i, (torch.Tensor, np.ndarray)) and i is Ellipsis]) > 1:
        raise SyntaxError
    dim, out = 0, self
    while index:
        item = index.pop(0)
        if isinstance(item, (list, tuple)):
            item = torch.tensor(item, device=self.device())
        if isinstance(item, np.ndarray):
            item = torch.from_numpy(item).to(self.device())
        if isinstance(item, int):
            out = out.select(dim, item)
            dim += 1
        elif isinstance(item, slice):
            if item.step is not None:
                raise ValueError("Step parameter not yet supported.")
            start = 0 if item.start is None else item.start
            start = self.size(dim) + start if start < 0 else start
            stop = self.size(dim) if item.stop is None else item.stop
            stop = self.size(dim) + stop if stop < 0 else stop
            out = out.narrow(dim, start, max(stop - start, 0))
            dim += 1
        elif torch.is_tensor(item):
            if item.dtype == torch.bool:
                out = out.masked_select(dim, item)
                dim += 1
            elif item.dtype == torch.int64:
                out = out.index_select(dim, item)
                dim += 1
        elif item is Ellipsis:
            if self.dim() - len(index) < dim:
                raise SyntaxError
            dim = self.dim() - len(index)
        else:
            raise SyntaxError
    return out


SparseTensor.narrow = lambda self, dim, start, length: narrow(self, dim, start, length)
SparseTensor.select = lambda self, dim, idx: select(self, dim, idx)
SparseTensor.index_select = lambda self, dim, idx: index_select(self, dim, idx)
SparseTensor.index_select_nnz = lambda self, idx, layout=None: index_select_nnz(self, idx, layout)
SparseTensor.masked_select = lambda self, dim, mask: masked_select(self, dim, mask)
SparseTensor.masked_select_nnz = lambda self, mask, layout=None: masked_select_nnz(self, mask, layout)
SparseTensor.__getitem__ = _getitem
