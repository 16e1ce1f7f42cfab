"""t() and transpose() — paddle_sparse/transpose.py:9-65.

`t()` is the cheap one: the CSC view of A *is* the CSR form of A^T, so with the
CSC caches in place (csr2csc, colptr, row[csr2csc] — one sort, storage.py)
only the values have to move.  `transpose()` is the functional form on raw
(index, value) and runs the coalesce chain on the swapped indices.
"""
from __future__ import annotations

import torch

from . import ops
from .coalesce import _coalesce_sorted_stream, _stack_index
from .storage import SparseStorage
from .tensor import SparseTensor

# cache of A  ->  the cache of A^T it becomes (transpose.py:24-29)
_SWAPPED = {"rowcount": "_colcount", "colptr": "_rowptr", "colcount": "_rowcount", "csr2csc": "_csc2csr"}


def t(src: SparseTensor) -> SparseTensor:
    st = src.storage
    to_csc = st.csr2csc()  # first: it leaves colptr and row[csr2csc] behind
    m, n = st.sparse_sizes()
    value = st.value()
    fields = {name: getattr(st, attr) for name, attr in _SWAPPED.items()}
    fields.update(
        # A^T's row index is col[csr2csc], the sorted column index.  With colptr
        # at hand it stays implicit: row() of the result expands it from rowptr
        # on first use, as for any CSR-built storage.
        row=None if st._colptr is not None else ops.gather_rows(st.col(), to_csc),
        rowptr=st._colptr,
        col=st._row_in_csc_order(),
        value=st._value_in_csc_order(),
        csc2csr=to_csc,
        sparse_sizes=(n, m),
    )
    # a permutation of an already validated storage: no re-sort, no range checks
    return src.from_storage(SparseStorage(is_sorted=True, trust_data=True, **fields))


SparseTensor.t = lambda self: t(self)


def transpose(index: torch.Tensor, value, m: int, n: int, coalesced: bool = True):
    """(index, value) of an m x n matrix -> those of its n x m transpose; with
    coalesced=True (the default) sorted row-major and duplicates added."""
    new_row, new_col = index[1].contiguous(), index[0].contiguous()
    if coalesced:
        new_row, new_col, value = _coalesce_sorted_stream(new_row, new_col, value, n, m, "add")
    return _stack_index(new_row, new_col), value
