"""SparseStorage — COO/CSR/CSC state with lazy caches, on MI355X HBM.

Same constructor, accessors and cache semantics as the reference class
(paddle_sparse/storage.py:31-776); the attribute names (`_row`, `_rowptr`,
`_col`, `_value`, `_rowcount`, `_colptr`, `_colcount`, `_csr2csc`, `_csc2csr`)
are kept because the reference's tests read them.  Every nnz-sized pass on
the hot path goes to a HIP kernel through paddle_sparse_amd.ops:

  ctor sort (storage.py:158-171) ... make_keys + sort_pairs | index_sort + split_keys
  row()/rowptr() (195-222) ......... ptr2ind / ind2ptr
  colcount()/colptr() (386-420) .... bincount / count2ptr (or ind2ptr)
  csr2csc()/csc2csr() (425-447) .... stable index_sort of col alone (+ ind2ptr of the
                                     sorted columns = colptr) / invert_permutation
  is_coalesced()/coalesce() (449-486) unique_sorted + segment_csr

Tensors are torch tensors; torch supplies allocation, views and O(1) glue
(`rowptr[1:] - rowptr[:-1]`), as Paddle does in the reference.
"""
from __future__ import annotations

import warnings
from typing import Callable, List, Optional, Tuple

import torch

from . import ops
from .utils import index_sort, is_pinned_tensor

layouts: List[str] = ["coo", "csr", "csc"]

_CACHES = ("rowcount", "colptr", "colcount", "csr2csc", "csc2csr")
_FIELDS = ("row", "rowptr", "col", "value") + _CACHES
HOT_COLUMNS = 65_536  # rows of the dense operand kept in the compact hot copy (32 MB at K = 128)
_SORT_BEATS_ATOMICS = 1 << 20  # entries from which column counts / sums go through the CSC order


def get_layout(layout: Optional[str] = None) -> str:
    if layout is None:
        layout = "coo"
        warnings.warn('`layout` argument unset, using default layout "coo". '
                      "This may lead to unexpected behaviour.")
    assert layout in layouts
    return layout


def _check_index(t: torch.Tensor, like: torch.Tensor, numel: Optional[int], what: str) -> torch.Tensor:
    assert t.dtype == torch.int64, f"{what} must be int64"
    assert t.device == like.device, f"{what} must live on {like.device}"
    assert t.dim() == 1, f"{what} must be 1-D"
    if numel is not None:
        assert t.numel() == numel, f"{what} has {t.numel()} entries, expected {numel}"
    return t.contiguous()


class SparseStorage(object):
    def __init__(
        self,
        row: Optional[torch.Tensor] = None,
        rowptr: Optional[torch.Tensor] = None,
        col: Optional[torch.Tensor] = None,
        value: Optional[torch.Tensor] = None,
        sparse_sizes: Optional[Tuple[Optional[int], Optional[int]]] = None,
        rowcount: Optional[torch.Tensor] = None,
        colptr: Optional[torch.Tensor] = None,
        colcount: Optional[torch.Tensor] = None,
        csr2csc: Optional[torch.Tensor] = None,
        csc2csr: Optional[torch.Tensor] = None,
        is_sorted: bool = False,
        trust_data: bool = False,
    ):
        assert row is not None or rowptr is not None
        assert col is not None
        col = _check_index(col, col, None, "col")
        nnz = col.numel()

        # storage.py:65-91 — infer (M, N) when not given; validate otherwise
        given_m = sparse_sizes[0] if sparse_sizes is not None else None
        given_n = sparse_sizes[1] if sparse_sizes is not None else None
        if given_m is None:
            if rowptr is not None:
                M = rowptr.numel() - 1
            else:
                M = int(row.max()) + 1 if row.numel() > 0 else 0
        else:
            M = int(given_m)
            if rowptr is not None:
                assert rowptr.numel() - 1 == M
            elif row.numel() > 0:
                assert trust_data or int(row.max()) < M
        if given_n is None:
            N = int(col.max()) + 1 if nnz > 0 else 0
        else:
            N = int(given_n)
            if nnz > 0:
                assert trust_data or int(col.max()) < N

        if row is not None:
            row = _check_index(row, col, nnz, "row")
        if rowptr is not None:
            rowptr = _check_index(rowptr, col, M + 1, "rowptr")
        if value is not None:
            assert value.device == col.device
            assert value.shape[0] == nnz
            value = value.contiguous()
        if rowcount is not None:
            rowcount = _check_index(rowcount, col, M, "rowcount")
        if colptr is not None:
            colptr = _check_index(colptr, col, N + 1, "colptr")
        if colcount is not None:
            colcount = _check_index(colcount, col, N, "colcount")
        if csr2csc is not None:
            csr2csc = _check_index(csr2csc, col, nnz, "csr2csc")
        if csc2csr is not None:
            csc2csr = _check_index(csc2csr, col, nnz, "csc2csr")

        self._row, self._rowptr, self._col, self._value = row, rowptr, col, value
        self._sparse_sizes: Tuple[int, int] = (M, N)
        self._rowcount, self._colptr, self._colcount = rowcount, colptr, colcount
        self._csr2csc, self._csc2csr = csr2csc, csc2csr
        self._row_csc: Optional[torch.Tensor] = None  # row[csr2csc], private
        self._edge_tags: Optional[torch.Tensor] = None  # uint8 per CSC edge, private (min/max backward)
        self._max_rowcount: Optional[int] = None  # longest row, private (structure only; one host read)
        self._spmm_algo_memo: Optional[str] = None  # SpMM forward kernel family for this row structure
        self._value_csc_memo = None  # (value, version, value[csr2csc]): see _value_in_csc_order
        self._hot_memo = None  # False | (hot column ids, col redirected into the compact copy): see _hot_columns
        self._csc_view_memo = None  # SparseStorage of the transpose over the CSC caches: see _csc_view
        self._perm_plans = {}  # "to_csr" / "to_csc": planned routes of a 4-byte array between the two orders

        # storage.py:158-171 — sort by (row, col) unless told it is sorted
        if not is_sorted and nnz > 0:
            keys, unsorted = ops.make_keys(self.row(), self._col, N, check_sorted=True)
            if int(unsorted.item()):
                # row[perm] / col[perm] (storage.py:166-167) are read back
                # from the sorted keys; a 4-byte scalar value rides the sort.
                if (value is not None and value.dim() == 1 and value.element_size() == 4
                        and not ops.needs_grad(value)):  # tracked values take the differentiable gather below
                    keys, self._value = ops.sort_pairs(keys, value, M * N, check=True)
                else:
                    keys, perm = index_sort(keys, M * N, with_sorted_inputs=True)
                    if value is not None:
                        self._value = ops.gather_rows(value, perm)
                self._row, self._col = ops.split_keys(keys, N)
                self._csr2csc = None
                self._csc2csr = None

    @classmethod
    def empty(cls):
        z = torch.empty(0, dtype=torch.int64)
        return cls(row=z, col=z.clone(), sparse_sizes=(0, 0), is_sorted=True, trust_data=True)

    # ---- canonical state --------------------------------------------------
    def has_row(self) -> bool:
        return self._row is not None

    def row(self) -> torch.Tensor:
        if self._row is None:
            if self._rowptr is None:
                raise ValueError
            self._row = ops.ptr2ind(self._rowptr, self._col.numel())  # storage.py:202
        return self._row

    def has_rowptr(self) -> bool:
        return self._rowptr is not None

    def rowptr(self) -> torch.Tensor:
        if self._rowptr is None:
            if self._row is None:
                raise ValueError
            self._rowptr = ops.ind2ptr(self._row, self._sparse_sizes[0])  # storage.py:218
        return self._rowptr

    def col(self) -> torch.Tensor:
        return self._col

    def has_value(self) -> bool:
        return self._value is not None

    def value(self) -> Optional[torch.Tensor]:
        return self._value

    def _prepare_value(self, value, layout):
        if value is not None:
            if get_layout(layout) == "csc":  # storage.py:246-247
                value = ops.gather_rows(value, self.csc2csr())
            value = value.contiguous()
            assert value.device == self._col.device
            assert value.shape[0] == self._col.numel()
        return value

    def set_value_(self, value: Optional[torch.Tensor], layout: Optional[str] = None):
        self._value = self._prepare_value(value, layout)
        return self

    def set_value(self, value: Optional[torch.Tensor], layout: Optional[str] = None):
        return self._replace(value=self._prepare_value(value, layout))

    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sparse_sizes

    def sparse_size(self, dim: int) -> int:
        return self._sparse_sizes[dim]

    def sparse_resize(self, sparse_sizes: Tuple[int, int]):
        """storage.py:281-331: grow/shrink the pointer and count caches."""
        assert len(sparse_sizes) == 2
        nnz = self._col.numel()

        def fit(ptr, cnt, diff):
            if diff > 0:
                if ptr is not None:
                    ptr = torch.cat([ptr, ptr.new_full((diff,), nnz)])
                if cnt is not None:
                    cnt = torch.cat([cnt, cnt.new_zeros(diff)])
            elif diff < 0:
                ptr = ptr[:diff] if ptr is not None else None
                cnt = cnt[:diff] if cnt is not None else None
            return ptr, cnt

        rowptr, rowcount = fit(self._rowptr, self._rowcount, sparse_sizes[0] - self._sparse_sizes[0])
        colptr, colcount = fit(self._colptr, self._colcount, sparse_sizes[1] - self._sparse_sizes[1])
        return self._replace(rowptr=rowptr, rowcount=rowcount, colptr=colptr,
                             colcount=colcount, sparse_sizes=tuple(sparse_sizes))

    def sparse_reshape(self, num_rows: int, num_cols: int):
        """storage.py:333-371: re-split the linear index row*N + col."""
        assert num_rows > 0 or num_rows == -1
        assert num_cols > 0 or num_cols == -1
        assert num_rows > 0 or num_cols > 0
        total = self._sparse_sizes[0] * self._sparse_sizes[1]
        if num_rows == -1:
            num_rows = total // num_cols
        if num_cols == -1:
            num_cols = total // num_rows
        assert num_rows * num_cols == total
        idx = self._sparse_sizes[1] * self.row() + self._col
        row = torch.div(idx, num_cols, rounding_mode="floor")
        col = idx - row * num_cols
        return SparseStorage(row=row, col=col, value=self._value,
                             sparse_sizes=(num_rows, num_cols), is_sorted=True, trust_data=True)

    # ---- lazy caches --------------------------------------------------------
    def has_rowcount(self) -> bool:
        return self._rowcount is not None

    def rowcount(self) -> torch.Tensor:
        if self._rowcount is None:
            p = self.rowptr()
            self._rowcount = p[1:] - p[:-1]  # storage.py:378-379
        return self._rowcount

    def has_colptr(self) -> bool:
        return self._colptr is not None

    def colptr(self) -> torch.Tensor:
        if self._colptr is None:
            if self._csr2csc is not None:  # storage.py:391-395
                self._colptr = ops.ind2ptr(ops.gather_rows(self._col, self._csr2csc),
                                           self._sparse_sizes[1])
            elif self._colcount is None and self._col.numel() >= _SORT_BEATS_ATOMICS:
                # counting columns with one device atomic per entry runs at ~20 G/s
                # on this chip (0.86 ms at 20 M entries); the 3-pass stable sort of
                # col takes 0.52 ms, gives colptr exactly and leaves csr2csc behind
                self.csr2csc()
            else:  # storage.py:397-398
                self._colptr = ops.count2ptr(self.colcount())
        return self._colptr

    def has_colcount(self) -> bool:
        return self._colcount is not None

    def colcount(self) -> torch.Tensor:
        if self._colcount is None:
            if self._colptr is not None:
                self._colcount = self._colptr[1:] - self._colptr[:-1]
            elif self._col.numel() >= _SORT_BEATS_ATOMICS:
                p = self.colptr()  # through the column sort, see colptr()
                self._colcount = p[1:] - p[:-1]
            else:  # storage.py:414-418 scatter_add(ones, col)
                self._colcount = ops.bincount(self._col, self._sparse_sizes[1])
        return self._colcount

    def has_csr2csc(self) -> bool:
        return self._csr2csc is not None

    def csr2csc(self) -> torch.Tensor:
        if self._csr2csc is None:
            # The reference sorts the key M*col + row (storage.py:430-432) because
            # its argsort is not stable.  The entries of a storage are in (row,
            # col) order already and the radix sort here IS stable, so ordering
            # them by col alone leaves equal columns in row order: the same
            # permutation from half the key bits, i.e. half the radix passes.
            N = self._sparse_sizes[1]
            col_csc, self._csr2csc = index_sort(self._col, N, with_sorted_inputs=True)
            # colptr = ind2ptr(col[csr2csc]) (storage.py:391-395) comes from the
            # sorted columns directly, no permutation gather
            if self._colptr is None and col_csc.numel() > 0:
                self._colptr = ops.ind2ptr(col_csc, N)
        return self._csr2csc

    def has_csc2csr(self) -> bool:
        return self._csc2csr is not None

    def csc2csr(self) -> torch.Tensor:
        if self._csc2csr is None:
            # storage.py:444-445 sorts the permutation; its inverse is the
            # same array and costs one scatter pass.
            self._csc2csr = ops.invert_permutation(self.csr2csc())
        return self._csc2csr

    def _row_in_csc_order(self) -> torch.Tensor:
        """row[csr2csc] (tensor.py:254-257), memoised for repeated backward."""
        if self._row_csc is None:
            self._row_csc = ops.gather_rows(self.row(), self.csr2csc())
        return self._row_csc

    def _csc_view(self):
        """The CSC view of this matrix as the CSR storage of its transpose (structure only,
        memoised): rowptr = colptr, col = row[csr2csc], colcount = rowcount.  The backward of
        SpMM wrt the dense operand is a forward SpMM over it (matmul.py), and it brings its own
        per-matrix choices with it — a power-law matrix has a power-law transpose, whose hub
        columns are this matrix's hub ROWS."""
        if getattr(self, "_csc_view_memo", None) is None:
            self.csr2csc()  # leaves colptr and row[csr2csc] behind
            M, N = self._sparse_sizes
            self._csc_view_memo = SparseStorage(row=None, rowptr=self.colptr(), col=self._row_in_csc_order(),
                                                value=None, sparse_sizes=(N, M), colcount=self.rowcount(),
                                                is_sorted=True, trust_data=True)
        return self._csc_view_memo

    def _permute_plan(self, direction: str, force: bool = False):
        """Planned route (ops.PermutePlan, structure only, memoised) of an nnz-sized 4-byte array
        between CSR and CSC order — "to_csc": out[j] = src[csr2csc[j]] (value[csr2csc],
        tensor.py:254-257), "to_csr": out[i] = src[csc2csr[i]] (grad_value's way back from the
        pass over the CSC view).  None below ops.PERMUTE_PLAN_FROM entries, where the plain gather
        is as fast (everything is cache resident) and the plan's 8 bytes per entry buy nothing —
        and None the FIRST time a direction is asked for: building a plan costs two sorts (~2 ms at
        20 M entries, five plain gathers' worth), which a one-off `t()` or `sum(dim=0)` never earns
        back; the second request (a training loop's second step) builds it.  force: build now."""
        nnz = self._col.numel()
        if nnz < ops.PERMUTE_PLAN_FROM or nnz >= (1 << 31):
            return None
        plans = getattr(self, "_perm_plans", None)
        if plans is None:
            plans = self._perm_plans = {}
        if direction not in plans:
            seen = plans.get(("asked", direction), 0)
            if not force and seen < 1:
                plans[("asked", direction)] = seen + 1
                return None
            # dest = the inverse of the gather's index array
            plans[direction] = ops.permute_plan(self.csc2csr() if direction == "to_csc" else self.csr2csc())
        return plans[direction]

    def _mean_scale_per_entry(self) -> torch.Tensor:
        """1 / max(deg(row), 1) of every entry, CSR order (fp32[nnz], structure only, memoised): with it the mean
        backward folds the scale into the weights BEFORE their planned way to CSC order and into grad_value AFTER
        its way back — two streaming multiplies instead of a dependent 4-byte read row_scale[r] per CSC entry
        inside the pass (+0.18 ms at 20 M entries)."""
        if getattr(self, "_mean_scale_memo", None) is None:
            scale = 1.0 / self.rowcount().clamp(min=1).to(torch.float32)
            self._mean_scale_memo = ops.gather_rows(scale, self.row())
        return self._mean_scale_memo

    def _longest_row(self) -> int:
        """Entries of the longest row (memoised; the min/max forward asks whether
        the one-byte form of arg_out is complete, i.e. no row above 128)."""
        if self._max_rowcount is None:
            self._max_rowcount = int(self.rowcount().max().item()) if self._sparse_sizes[0] > 0 else 0
        return self._max_rowcount

    def _value_in_csc_order(self) -> Optional[torch.Tensor]:
        """value[csr2csc] (tensor.py:254-257, transpose.py:19-22, the dim-0 reduction).
        Kept with the storage for as long as the value tensor is the same object at
        the same version — a fixed adjacency asks for it every step (a 4-byte random
        gather: 0.5 ms at 20 M entries, against 0.05 ms for the reduction that reads
        it).  Values that autograd tracks are gathered afresh through the
        differentiable ops.gather_rows."""
        value = self._value
        if value is None:
            return None
        perm = self.csr2csc()
        if ops.needs_grad(value):
            return ops.gather_rows(value, perm, inverse=self.csc2csr())
        memo = self._value_csc_memo
        if (memo is not None and memo[0] is value and memo[1] == value._version
                and memo[2]._version == memo[3]):  # neither side written in place since
            return memo[2]
        plan = self._permute_plan("to_csc") if (value.dim() == 1 and value.element_size() == 4) else None
        out = ops.permute_apply(value.contiguous(), plan) if plan is not None else ops.gather_rows(value, perm)
        self._value_csc_memo = (value, value._version, out, out._version)
        return out

    def _hot_columns(self):
        """Hub columns of a power-law matrix, for the SpMM forward's compact copy of the hot
        rows of the dense operand (psa_spmm_coo, hot_rows).  Returns None, or (hot, col_eff):
        hot = ids of the HOT_COLUMNS most referenced columns (by count), col_eff = col with
        every reference to hot[j] replaced by N + j.  Structure only, memoised; built when the
        matrix takes the edge-range forward and the hot columns draw at least a fifth of all
        entries and four times their uniform share (on a uniform 2 M-column graph the 65 536
        most referenced columns draw 3 %: nothing to gain).  Two host reads, once."""
        if self._hot_memo is None:
            self._hot_memo = False
            N, nnz = self._sparse_sizes[1], self._col.numel()
            k = min(HOT_COLUMNS, N // 4)
            if self._spmm_algo() == "edge_ranges" and k >= 64 and nnz > 0 and N + k < (1 << 31):
                count = self.colcount()
                top = int(count.max().item())
                # stable sort of (top - count): the most referenced columns first, ties by id
                _, order = index_sort(top - count, top + 1)
                hot = order[:k].contiguous()
                drawn = int(ops.gather_rows(count, hot).sum().item())
                if drawn * 5 >= nnz and drawn * N >= 4 * k * nnz:  # a fifth of all entries, 4x their uniform share
                    slot = torch.full((N,), -1, dtype=torch.int64, device=hot.device)
                    slot[hot] = torch.arange(k, dtype=torch.int64, device=hot.device)
                    s = ops.gather_rows(slot, self._col)
                    self._hot_memo = (hot, torch.where(s >= 0, s + N, self._col).contiguous())
        return self._hot_memo or None

    def _huge_rows(self):
        """Rows above ops.ARG_WORDS_EXACT_ROW (65 535) entries, cut into PIECES of at most that many — the form in
        which the min / max training step keeps the row-local arg_out exact on such rows without an int64 arg_out
        (matmul.py): every piece is a row of a small CSR matrix of its own (`rowptr`, `col`, entry ids `ids` into this
        storage's arrays), reduced beside the main product; the piece that wins a (row, k) keeps its two-byte
        winner, the others say "no winner".  None when no row is that long.  Structure only, memoised.
        Fields: rows int64[H] (ids, ascending), start int64[H] (rowptr of each), piece_ptr int64[H + 1] (pieces of
        each), piece_row int64[P] (position in `rows` of each piece), rowptr int64[P + 1], col int64[E], ids
        int64[E]."""
        memo = getattr(self, "_huge_memo", None)
        if memo is None:
            memo = False
            if self._longest_row() > ops.ARG_WORDS_EXACT_ROW:
                cut = ops.ARG_WORDS_EXACT_ROW
                count, rowptr = self.rowcount(), self.rowptr()
                rows = torch.nonzero(count > cut).flatten()
                deg, start = count[rows], rowptr[rows]
                dev = rows.device
                H = rows.numel()
                per = (deg + cut - 1) // cut
                piece_ptr = torch.zeros(H + 1, dtype=torch.int64, device=dev)
                piece_ptr[1:] = torch.cumsum(per, 0)
                P = int(piece_ptr[-1].item())
                piece_row = torch.repeat_interleave(torch.arange(H, device=dev), per, output_size=P)
                piece_k = torch.arange(P, device=dev) - piece_ptr[piece_row]
                piece_len = torch.clamp(deg[piece_row] - piece_k * cut, max=cut)
                rp = torch.zeros(P + 1, dtype=torch.int64, device=dev)
                rp[1:] = torch.cumsum(piece_len, 0)
                E = int(rp[-1].item())
                ent_slot = torch.repeat_interleave(torch.arange(H, device=dev), deg, output_size=E)
                first = torch.cumsum(deg, 0) - deg
                ids = (start[ent_slot] + (torch.arange(E, device=dev) - first[ent_slot])).contiguous()
                memo = dict(rows=rows, start=start, piece_ptr=piece_ptr, piece_row=piece_row, rowptr=rp,
                            col=ops.gather_rows(self._col, ids), ids=ids)
            self._huge_memo = memo
        return memo or None

    def _huge_backward_plan(self):
        """The pass over the CSC view for a matrix with rows above 65 535 entries (`_huge_rows`): every CSC entry
        of such a row is pointed at its PIECE — a row of the compact copies the pass already reads hub rows from
        (ids >= M name rows of `hot_grad` / `hot_bytes`; the pieces follow the view's own hub rows there) — and
        tagged with its position inside the piece.  Returns (hot_ids int64[h + P]: the rows of grad_out to copy,
        row_csc int64[nnz], tags int16[nnz], P).  Structure only, memoised."""
        memo = getattr(self, "_huge_bw_memo", None)
        if memo is None:
            hr = self._huge_rows()
            M = self._sparse_sizes[0]
            cut = ops.ARG_WORDS_EXACT_ROW
            base = self._csc_view()._hot_columns()
            hot_ids = base[0] if base is not None else torch.empty(0, dtype=torch.int64, device=self._col.device)
            row_true = self._row_in_csc_order()
            row_eff = (base[1] if base is not None else row_true).clone()
            tags = self._csc_edge_tags(2).clone()
            slot_of_row = torch.full((M,), -1, dtype=torch.int64, device=row_true.device)
            slot_of_row[hr["rows"]] = torch.arange(hr["rows"].numel(), device=row_true.device)
            slot = ops.gather_rows(slot_of_row, row_true)
            j = torch.nonzero(slot >= 0).flatten()  # CSC positions of the entries of huge rows
            sj = slot[j]
            local = self.csr2csc()[j] - hr["start"][sj]
            row_eff[j] = M + hot_ids.numel() + hr["piece_ptr"][sj] + torch.div(local, cut, rounding_mode="floor")
            within = local % cut  # 0 .. 65 534, stored as the two-byte pattern
            tags[j] = torch.where(within >= 32768, within - 65536, within).to(torch.int16)
            ids_ext = torch.cat([hot_ids, hr["rows"][hr["piece_row"]]]).contiguous()
            memo = self._huge_bw_memo = (ids_ext, row_eff.contiguous(), tags.contiguous(), int(hr["piece_row"].numel()))
        return memo

    def _spmm_algo(self) -> str:
        """Which SpMM forward suits this row structure (memoised; one 32-byte host
        read per matrix): "edge_ranges" once rows with at most two entries — the
        empty ones included — make up more than 40 % of the rows (power-law
        graphs), else "row_waves".  See psa_csr_row_stats in the header."""
        if self._spmm_algo_memo is None:
            M = self._sparse_sizes[0]
            empty, tiny, _, longest = ops.csr_row_stats(self.rowptr()) if M > 0 else (0, 0, 0, 0)
            self._max_rowcount = longest
            fits = M < (1 << 31) and self._sparse_sizes[1] < (1 << 31) and self._col.numel() < (1 << 31)  # 31-bit ids in the edge-range kernels
            self._spmm_algo_memo = "edge_ranges" if fits and 5 * (empty + tiny) > 2 * M else "row_waves"
        return self._spmm_algo_memo

    def _csc_edge_tags(self, width: int = 1) -> torch.Tensor:
        """Position of every CSC-ordered edge inside its CSR row, `width` bytes each
        (ops.csc_edge_tags): structure only, memoised per width for the min/max backward."""
        if self._edge_tags is None:
            self._edge_tags = {}
        if width not in self._edge_tags:
            self._edge_tags[width] = ops.csc_edge_tags(self.rowptr(), self._row_in_csc_order(), self.csr2csc(), width)
        return self._edge_tags[width]

    # ---- coalesce -------------------------------------------------------------
    def is_coalesced(self) -> bool:
        """storage.py:449-452: keys strictly increasing."""
        nnz = self._col.numel()
        if nnz == 0:
            return True
        keys, unsorted = ops.make_keys(self.row(), self._col, self._sparse_sizes[1], check_sorted=True)
        count, _, _, _ = ops.unique_sorted(keys, 1, want_ptr=False, want_rowcol=False)
        return count == nnz and not int(unsorted.item())

    def coalesce(self, reduce: str = "add"):
        """storage.py:454-486."""
        nnz = self._col.numel()
        if nnz == 0:
            return self
        N = self._sparse_sizes[1]
        keys, _ = ops.make_keys(self.row(), self._col, N)
        value = self._value
        if (value is not None and value.dim() == 1 and value.dtype in (torch.float32, torch.int32)
                and not ops.needs_grad(value)):
            # 4-byte scalar values of a sorted storage: index and reduced values from one launch, no ptr array
            count, row, col, value = ops.unique_sorted_reduce(keys, N, value, reduce)
            if count == nnz:
                return self
        else:
            count, ptr, row, col = ops.unique_sorted(keys, N)
            if count == nnz:  # already coalesced (storage.py:459)
                return self
            if value is not None:
                value = ops.segment_csr(value, ptr, reduce)
        return SparseStorage(row=row, col=col, value=value, sparse_sizes=self._sparse_sizes,
                             is_sorted=True, trust_data=True)

    def fill_cache_(self):
        self.row()
        self.rowptr()
        self.rowcount()
        self.colptr()
        self.colcount()
        self.csr2csc()
        self.csc2csr()
        return self

    def clear_cache_(self):
        for k in _CACHES:
            setattr(self, "_" + k, None)
        self._row_csc = None
        self._edge_tags = None
        self._max_rowcount = None
        self._spmm_algo_memo = None
        self._value_csc_memo = None
        self._hot_memo = None
        self._csc_view_memo = None
        self._perm_plans = {}
        self._mean_scale_memo = None
        self._huge_memo = self._huge_bw_memo = None
        return self

    def cached_keys(self) -> List[str]:
        return [k for k in _CACHES if getattr(self, "_" + k) is not None]

    def num_cached_keys(self) -> int:
        return len(self.cached_keys())

    # ---- copies / moves ----------------------------------------------------------
    def _replace(self, **kw):
        """New storage sharing every field not named in kw (no re-sort)."""
        args = {k: getattr(self, "_" + k) for k in _FIELDS}
        args["sparse_sizes"] = self._sparse_sizes
        args.update(kw)
        out = SparseStorage(is_sorted=True, trust_data=True, **args)
        if not (set(kw) - {"value"}):  # same sparsity structure: the private CSC helpers carry over
            out._row_csc, out._edge_tags = self._row_csc, self._edge_tags
            out._max_rowcount = self._max_rowcount
            out._spmm_algo_memo = self._spmm_algo_memo
            out._hot_memo = self._hot_memo
            out._csc_view_memo = self._csc_view_memo
            out._perm_plans = self._perm_plans
            out._mean_scale_memo = getattr(self, "_mean_scale_memo", None)
            out._huge_memo, out._huge_bw_memo = getattr(self, "_huge_memo", None), getattr(self, "_huge_bw_memo", None)
        return out

    def _map(self, fn: Callable[[torch.Tensor], torch.Tensor]):
        return self._replace(**{k: (None if getattr(self, "_" + k) is None else fn(getattr(self, "_" + k)))
                                for k in _FIELDS})

    def copy(self):
        return self._replace()

    def clone(self):
        return self._map(lambda t: t.clone())

    def type(self, dtype: torch.dtype, non_blocking: bool = False):
        value = self._value
        if value is None or dtype == value.dtype:
            return self
        return self.set_value(value.to(dtype=dtype, non_blocking=non_blocking), layout="coo")

    def type_as(self, tensor: torch.Tensor, non_blocking: bool = False):
        return self.type(dtype=tensor.dtype, non_blocking=non_blocking)

    def to_device(self, device, non_blocking: bool = False):
        device = torch.device(device)
        if device == self._col.device:
            return self
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def device_as(self, tensor: torch.Tensor, non_blocking: bool = False):
        return self.to_device(device=tensor.device, non_blocking=non_blocking)

    def cuda(self):
        if self._col.is_cuda:
            return self
        return self._map(lambda t: t.cuda())

    def pin_memory(self):
        return self._map(lambda t: t.pin_memory())

    def is_pinned(self) -> bool:
        return all(is_pinned_tensor(getattr(self, "_" + k)) for k in _FIELDS
                   if getattr(self, "_" + k) is not None)

    def share_memory_(self):
        for k in _FIELDS:
            t = getattr(self, "_" + k)
            if t is not None and not t.is_cuda:
                t.share_memory_()
        return self

    def is_shared(self) -> bool:
        return all(getattr(self, "_" + k).is_shared() for k in _FIELDS
                   if getattr(self, "_" + k) is not None)
