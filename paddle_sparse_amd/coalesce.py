"""coalesce(index, value, m, n, op) — paddle_sparse/coalesce.py:8-29.

The reference builds an unsorted SparseStorage (argsort + 3 gathers,
storage.py:158-171) and then coalesces it (storage.py:454-486).  The same
result is produced here in one fused chain that never materialises the
permuted row/col/value arrays:

    keys = row*n + col            make_keys (also tells if already sorted)
    sorted_keys, perm             index_sort (stable LSD radix, HIP)
    count, ptr, row', col'        unique_sorted (row'/col' = key / n, key % n)
    value' = reduce value[perm]   segment_csr(perm=perm)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops


_ONE_WORKGROUP_BELOW = 10_240  # entries the one-workgroup, LDS-resident sort of the chain takes
_CHAIN_BELOW = 1 << 20  # entries: the two-call chain (always sorts); above, asking whether the keys
# are sorted already is worth its host read (a sorted 100 M-entry input skips 4 ms of sort)


def _coalesce_sorted_stream(row, col, value, m: int, n: int, op: str):
    nnz = col.numel()
    if nnz == 0:
        return row, col, value
    if (nnz <= _CHAIN_BELOW and m > 0 and n > 0 and m * n < (1 << 62)
            and (value is None or value.dtype in ops._DTYPE_ID) and not ops.needs_grad(value)):
        # psa_coalesce_count + psa_coalesce_write on worst-case buffers, one host read at the end
        index, value, _ = ops.coalesce_chain(row, col, value, m, n, op)
        return index[0], index[1], value
    keys, status = ops.make_keys_checked(row, col, m, n)
    flags = int(status[1].item())
    if flags & 1:
        raise ops.IndexRangeError(f"coalesce: an index lies outside the {m} x {n} matrix")
    perm = scratch = None
    was_sorted = not (flags & 2)
    if not was_sorted:
        if value is not None and value.dim() == 1 and value.element_size() == 4 and not ops.needs_grad(value):
            # 4-byte scalar values ride through the sort as the payload: the
            # reduce below then reads them as a stream, not as value[perm[i]]
            # (values that autograd tracks go through the permutation instead:
            # ops.segment_csr / ops.gather_rows are differentiable)
            keys, value, scratch = ops.sort_pairs(keys, value, m * n, keep_scratch=True)
        else:
            keys, perm, scratch = ops.index_sort(keys, m * n, with_sorted_inputs=True, keep_scratch=True)
    if (value is not None and perm is None and value.dim() == 1 and value.dtype in (torch.float32, torch.int32)
            and not ops.needs_grad(value)):
        # values in sorted order (they rode the sort, or the input was sorted): index and reduced values from one
        # launch when the runs are short — no ptr array (24 -> 16 bytes written per distinct entry, 8 fewer read)
        count, new_row, new_col, value_out = ops.unique_sorted_reduce(keys, n, value, op, after=scratch)
        if count == nnz and was_sorted:
            return row, col, value
        return new_row, new_col, value_out
    # the count read below also brings back the sort's look-back diagnostic
    count, ptr, new_row, new_col = ops.unique_sorted(keys, n, after=scratch)
    if count == nnz and was_sorted:
        return row, col, value  # sorted and duplicate-free already
    if value is not None:
        if count < nnz:
            value = ops.segment_csr(value, ptr, op, perm=perm)
        elif perm is not None:
            value = ops.gather_rows(value, perm)
    return new_row, new_col, value


def _coalesce_two_sorted(row_a, col_a, value_a, row_b, col_b, value_b, n: int, op: str):
    """Coalesce of the concatenation [A; B] when A and B are each in (row, col)
    order already (add.py:30-47, tensor.py:415-451): a stable merge of the two
    key streams (ops.merge_sorted, two streaming launches) stands in for the
    radix sort; run lengths and the segmented reduce are the usual ones, and
    the terms of every entry keep the order a stable sort would give them.
    The result has values only when both operands do."""
    keys_a, _ = ops.make_keys(row_a, col_a, n)
    keys_b, _ = ops.make_keys(row_b, col_b, n)
    total = keys_a.numel() + keys_b.numel()
    has_value = value_a is not None and value_b is not None
    rides = (has_value and value_a.dim() == 1 and value_b.dim() == 1 and value_a.element_size() == 4
             and value_a.dtype == value_b.dtype and not ops.needs_grad(value_a) and not ops.needs_grad(value_b))
    keys, source, value = ops.merge_sorted(keys_a, keys_b, value_a.contiguous() if rides else None,
                                           value_b.contiguous() if rides else None,
                                           want_source=has_value and not rides)
    if rides and value.dtype in (torch.float32, torch.int32):
        # merged values are in key order already: index and reduced values from one launch, no ptr array
        _, row, col, value = ops.unique_sorted_reduce(keys, n, value, op)
        return row, col, value
    count, ptr, row, col = ops.unique_sorted(keys, n)
    if not has_value:
        return row, col, None
    if rides:
        return row, col, (ops.segment_csr(value, ptr, op) if count < total else value)
    both = torch.cat([value_a, value_b], dim=0)
    return row, col, (ops.segment_csr(both, ptr, op, perm=source) if count < total
                      else ops.gather_rows(both, source))


def _stack_index(row: torch.Tensor, col: torch.Tensor) -> torch.Tensor:
    """stack([row, col]) (coalesce.py:29) without the copy when col already
    follows row in one buffer (the coalesce chain and ops.unique_sorted write
    them that way)."""
    n = row.numel()
    if (row.dim() == 1 and col.dim() == 1 and col.numel() == n and row.is_contiguous() and col.is_contiguous()
            and row.untyped_storage().data_ptr() == col.untyped_storage().data_ptr()
            and col.storage_offset() == row.storage_offset() + n):
        return torch.as_strided(row, (2, n), (n, 1))
    return torch.stack([row, col], dim=0)


def coalesce(index: torch.Tensor, value: Optional[torch.Tensor], m: int, n: int,
             op: str = "add") -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Row-wise sorts `index` and merges duplicate entries with `op`
    ("add"/"sum", "mean", "min", "max").  Returns (index[2, nnz'], value)."""
    row, col = index[0].contiguous(), index[1].contiguous()
    assert row.dtype == torch.int64 and col.dtype == torch.int64
    nnz = col.numel()
    if value is not None:
        assert value.shape[0] == nnz
    if (0 < nnz <= _CHAIN_BELOW and m > 0 and n > 0 and m * n < (1 << 62)
            and (value is None or value.dtype in ops._DTYPE_ID) and not ops.needs_grad(value)):
        # the chain hands back the [2, nnz'] index as one buffer: nothing to split and restack
        # (config 1 is launch- and host-bound: every microsecond of Python shows)
        out_index, out_value, _ = ops.coalesce_chain(row, col, value, m, n, op)
        return out_index, out_value
    row, col, value = _coalesce_sorted_stream(row, col, value, m, n, op)
    return _stack_index(row, col), value
