"""spspmm(indexA, valueA, indexB, valueB, m, k, n) — sparse x sparse product
(README.md:308-353; documented by the reference, no kernel in its tree).

Expand / sort / compress on the path's own HIP kernels:

    counts[e]  = |B row colA[e]|                 spspmm_count
    offsets    = [0, cumsum(counts)]             count2ptr      (one host read: total)
    owner[p]   = A entry of product p            ptr2ind
    key, val   = (i*n + j, a*b) per product      spspmm_expand
    sort by key (stable), add the runs           the coalesce chain (coalesce.py)

With 4-byte values the walk runs over the CSC views instead (`_spspmm_by_column`:
outer = the entries of B in column order, inner = a column of A, key =
(i << 32) | j), so the products come out grouped by output column and ONE
stable sort on the row field of the key (3 radix passes at 2 M rows instead of
6 on i*n + j) puts them in (i, j) order: 11.6 ms instead of 14.7 ms for A @ A
on the config-3 graph, term order unchanged.

The products of one C entry reach the segmented sum in the order a sequential
row-by-row product meets them (A's storage order, then B's).  While runs
average fewer than 32 products the sum is taken in exactly that order and fp32
results equal a Gustavson CPU product bit for bit; longer runs are summed
lane-strided by one wave each (a fixed order too, within fp32 rounding of the
sequential one).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from .coalesce import _stack_index, coalesce


def _spspmm_csr(rowA, colA, valueA, rowptrB, colB, valueB, m: int, n: int):
    """A as sorted COO (row, col, value), B as CSR; returns (row, col, value) of C."""
    dtype = valueA.dtype if valueA is not None else (valueB.dtype if valueB is not None else torch.float32)
    device = colA.device
    empty_i = torch.empty(0, dtype=torch.int64, device=device)
    has_value = valueA is not None or valueB is not None
    empty_v = torch.empty(0, dtype=dtype, device=device) if has_value else None
    if colA.numel() == 0 or colB.numel() == 0:
        return empty_i, empty_i.clone(), empty_v
    counts = ops.spspmm_count(colA, rowptrB)
    offsets = ops.count2ptr(counts)
    total = int(offsets[-1].item())
    if total == 0:
        return empty_i, empty_i.clone(), empty_v
    owner = ops.ptr2ind(offsets, total)
    keys, vals = ops.spspmm_expand(rowA, colA, valueA, rowptrB, colB, valueB, offsets, owner,
                                   total, n, dtype)
    del owner, offsets, counts
    # rows of A are sorted, so keys are already grouped by C row; the sort
    # orders columns inside rows and brings equal (i, j) together
    if vals is not None and vals.element_size() == 4:
        keys, vals, scratch = ops.sort_pairs(keys, vals, m * n, keep_scratch=True)
        perm = None
    else:
        keys, perm, scratch = ops.index_sort(keys, m * n, with_sorted_inputs=True, keep_scratch=True)
    if vals is not None and perm is None and vals.dtype in (torch.float32, torch.int32):
        # products in key order: index and summed values from one launch, no ptr array
        _, row, col, vals = ops.unique_sorted_reduce(keys, n, vals, "sum", after=scratch)
        return row, col, vals
    count, ptr, row, col = ops.unique_sorted(keys, n, after=scratch)  # the sort's fault word rides the count
    if vals is not None:
        if count < total:
            vals = ops.segment_csr(vals, ptr, "sum", perm=perm)
        elif perm is not None:
            vals = ops.gather_rows(vals, perm)
    return row, col, vals


def _spspmm_by_column(a, b):
    """C = A @ B for SparseTensors with 4-byte scalar values, walking the CSC
    views: for every entry (c, j) of B, in column order, and every entry (i, c)
    of A's column c, emit ((i << 32) | j, a * b).  Products then come grouped by
    output column j, so ONE stable sort on the row field (3 radix passes for
    2 M rows, against 6 for row * n + col) gives (row, col) order; the terms of
    each entry keep Gustavson's order (c ascending)."""
    m, n = a.size(0), b.size(1)
    colptrA, rowA_csc, valA_csc = a.csc()
    colptrB, rowB_csc, valB_csc = b.csc()
    nnzB = rowB_csc.numel()
    dtype = valA_csc.dtype if valA_csc is not None else valB_csc.dtype
    empty_i = torch.empty(0, dtype=torch.int64, device=rowA_csc.device)
    if rowA_csc.numel() == 0 or nnzB == 0:
        return empty_i, empty_i.clone(), torch.empty(0, dtype=dtype, device=empty_i.device)
    counts = ops.spspmm_count(rowB_csc, colptrA)  # |column c of A| for every B entry (c, j)
    offsets = ops.count2ptr(counts)
    total = int(offsets[-1].item())
    if total == 0:
        return empty_i, empty_i.clone(), torch.empty(0, dtype=dtype, device=empty_i.device)
    owner = ops.ptr2ind(offsets, total)
    colB_of_entry = ops.ptr2ind(colptrB, nnzB)  # j of every CSC-ordered B entry
    keys, vals = ops.spspmm_expand(colB_of_entry, rowB_csc, valB_csc, colptrA, rowA_csc, valA_csc, offsets,
                                   owner, total, -1, dtype)
    del owner, offsets, counts
    keys, vals, scratch = ops.sort_pairs_field(keys, vals, 32, m, keep_scratch=True)
    if vals.dtype in (torch.float32, torch.int32):  # index and summed values from one launch, no ptr array
        _, row, col, vals = ops.unique_sorted_reduce(keys, 1 << 32, vals, "sum", after=scratch)
        return row, col, vals
    count, ptr, row, col = ops.unique_sorted(keys, 1 << 32, after=scratch)
    if count < total:
        vals = ops.segment_csr(vals, ptr, "sum")
    return row, col, vals


def _by_column_ok(valueA, valueB, m: int, n: int) -> bool:
    v = valueA if valueA is not None else valueB
    return (v is not None and v.dim() == 1 and v.element_size() == 4 and m < (1 << 31) and n < (1 << 31)
            and (valueA is None or valueB is None or valueA.dtype == valueB.dtype))


def spspmm(indexA: torch.Tensor, valueA: Optional[torch.Tensor], indexB: torch.Tensor,
           valueB: Optional[torch.Tensor], m: int, k: int, n: int, coalesced: bool = False
           ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Matrix product of two sparse matrices given as COO (index[2, nnz], value).
    Both must be coalesced (row-major sorted, no duplicates); `coalesced=True`
    coalesces them first.  Returns the coalesced (index, value) of the [m, n]
    product."""
    if coalesced:
        indexA, valueA = coalesce(indexA, valueA, m, k)
        indexB, valueB = coalesce(indexB, valueB, k, n)
    rowA, colA = indexA[0].contiguous(), indexA[1].contiguous()
    rowB, colB = indexB[0].contiguous(), indexB[1].contiguous()
    if _by_column_ok(valueA, valueB, m, n):
        from .tensor import SparseTensor

        a = SparseTensor(row=rowA, col=colA, value=valueA, sparse_sizes=(m, k), is_sorted=True, trust_data=True)
        b = SparseTensor(row=rowB, col=colB, value=valueB, sparse_sizes=(k, n), is_sorted=True, trust_data=True)
        row, col, value = _spspmm_by_column(a, b)
        return _stack_index(row, col), value
    rowptrB = ops.ind2ptr(rowB, k)
    row, col, value = _spspmm_csr(rowA, colA, valueA, rowptrB, colB, valueB, m, n)
    return _stack_index(row, col), value


def spspmm_tensor(a, b):
    """SparseTensor @ SparseTensor -> SparseTensor (the `matmul(src, other)`
    branch for a sparse `other`)."""
    from .tensor import SparseTensor

    assert a.size(1) == b.size(0), "inner dimensions differ"
    m, n = a.size(0), b.size(1)
    if _by_column_ok(a.storage.value(), b.storage.value(), m, n):
        row, col, value = _spspmm_by_column(a, b)
    else:
        rowA, colA, valueA = a.coo()
        rowptrB, colB, valueB = b.csr()
        row, col, value = _spspmm_csr(rowA, colA, valueA, rowptrB, colB, valueB, m, n)
    return SparseTensor(row=row, col=col, value=value, sparse_sizes=(m, n), is_sorted=True,
                        trust_data=True)


__all__ = ["spspmm", "spspmm_tensor"]
