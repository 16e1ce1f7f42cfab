"""t() and transpose() — paddle_sparse/transpose.py:9-65."""
from __future__ import annotations

import torch

from . import ops
from .coalesce import _coalesce_sorted_stream, _stack_index
from .storage import SparseStorage
from .tensor import SparseTensor


def t(src: SparseTensor) -> SparseTensor:
    """transpose.py:9-33: permute by csr2csc and swap the row/col caches."""
    st = src.storage
    csr2csc = st.csr2csc()
    row, col, value = src.coo()
    if value is not None:
        value = ops.gather_rows(value, csr2csc)
    M, N = st.sparse_sizes()
    # The transposed row index is col[csr2csc] = the sorted column index.  With
    # colptr cached (csr2csc() leaves it behind) it stays implicit: row() of the
    # result expands it from rowptr on first use, as for any CSR-built storage.
    storage = SparseStorage(
        row=ops.gather_rows(col, csr2csc) if st._colptr is None else None,
        rowptr=st._colptr,
        col=st._row_in_csc_order(),
        value=value,
        sparse_sizes=(N, M),
        rowcount=st._colcount,
        colptr=st._rowptr,
        colcount=st._rowcount,
        csr2csc=st._csc2csr,
        csc2csr=csr2csc,
        is_sorted=True,
        trust_data=True,  # a permutation of an already validated storage
    )
    return src.from_storage(storage)


SparseTensor.t = lambda self: t(self)


def transpose(index, value, m, n, coalesced=True):
    """transpose.py:41-65: swap rows and columns; with coalesced=True the
    result is sorted and duplicate-free (duplicates added)."""
    row, col = index[1].contiguous(), index[0].contiguous()
    if coalesced:
        row, col, value = _coalesce_sorted_stream(row, col, value, n, m, "add")
    return _stack_index(row, col), value
