"""SpMM on the SparseTensor surface: spmm / matmul / `@`, with autograd.

The reference lists these as "Support later" (README.md:47-50) and documents
the functional form `spmm(index, value, m, n, matrix)` (README.md:267-306).
Semantics and the backward formulas are upstream pytorch_sparse
(torch_sparse/matmul.py): sum/mean differentiate wrt the dense operand with
an SpMM over the CSC view and wrt the values with an SDDMM-shaped kernel;
min/max route gradients through arg_out.  All compute is HIP
(psa_spmm, psa_spmm_value_bw, psa_transpose_weights, psa_spmm_sum_bw_csc,
psa_spmm_minmax_bw_csc, psa_spmm_minmax_bw).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .coalesce import coalesce
from .storage import SparseStorage
from .tensor import SparseTensor


HUGE_IDS = 1 << 20  # room kept below 2^31 for the ids of hub rows and pieces in the compact copies
HUGE_ROW_PIECES = True  # test hook: False sends matrices with rows above 65 535 entries back to the int64 arg_out


def _csc_weights(st: SparseStorage, value: Optional[torch.Tensor], csr2csc, row_csc, mean: bool):
    """Edge weights in CSC order for grad_mat = A^T grad_out (value[csr2csc],
    over deg(row) for mean).  A fixed adjacency — the usual GNN case — asks for
    the same array every step, so the last one is kept on the storage together
    with the value tensor it was built from: holding that tensor keeps its
    address from being reused, and its version counter tells an in-place update."""
    memo = getattr(st, "_csc_weight_memo", None)
    if memo is not None:
        src, version, was_mean, w = memo
        same = (src is None and value is None) or (
            src is not None and value is not None and src.data_ptr() == value.data_ptr()
            and src.shape == value.shape and version == value._version)
        if same and was_mean == mean:
            return w
    w = None if mean else _streamed_values(st, value)  # value[csr2csc] alone: the planned route when there is one
    if w is None:
        w = ops.transpose_weights(value, csr2csc, row_csc, st.rowptr(), mean)
    st._csc_weight_memo = (value, None if value is None else value._version, mean, w)
    return w


def _streamed_values(st: SparseStorage, value: Optional[torch.Tensor], plan="ask") -> Optional[torch.Tensor]:
    """value[csr2csc] through the storage's planned route (two streaming passes, 0.12 ms at 20 M
    entries), for the passes over the CSC view: they then read their weights as a stream instead
    of through nnz dependent 4-byte reads value[csr2csc[j]] (18 % of such a pass's memory
    requests).  None when there is no plan (small matrices) — the pass reads through csr2csc."""
    if value is None or value.dtype != torch.float32 or value.dim() != 1:
        return None
    if isinstance(plan, str):  # "ask": one request to the storage (it builds a plan on the second one)
        plan = st._permute_plan("to_csc")
    return None if plan is None else ops.permute_apply(value.detach().contiguous(), plan)


def _half_minmax_bw_ok(st: SparseStorage, K: int) -> bool:
    """Will the half-width masked pass over the CSC view serve the min / max backward of this matrix?  K in one
    tile and an exact row-local form (no row above 65 535 entries).  Both forward families leave that form behind in
    half width, and the transpose may be anything: long columns run in chunks."""
    return ops.half_sum_bw_csc_supported(K) and st._longest_row() <= ops.ARG_WORDS_EXACT_ROW


def _huge_piece_winners(st: SparseStorage, reduce: str, value: Optional[torch.Tensor], mat: torch.Tensor) -> torch.Tensor:
    """min / max on a matrix with rows above 65 535 entries, without an int64 arg_out: the rows in question are
    reduced once more as PIECES of at most 65 535 entries (`SparseStorage._huge_rows`: a small CSR of its own, the
    same kernels), each piece leaving its exact two-byte row-local winners; per (row, k) the first piece that
    reaches the row's extreme keeps its winner, every other piece says "no winner" (0xffff) — ties go to the
    earlier edge, as everywhere.  Returns int16[P, K]: what the pass over the CSC view reads for the entries of
    those rows (`_huge_backward_plan` points them at their pieces).  The extra work is one more gather of the
    entries of the rows concerned; everything else is a few launches on [P, K] arrays.  (A row whose extreme is
    NaN gets no winner here; the int64 route would name one.)"""
    hr = st._huge_rows()
    v = None if value is None else ops.gather_rows(value.detach(), hr["ids"])
    out_h, _, bytes_h = ops._spmm(reduce, hr["rowptr"], hr["col"], v, mat.detach(), want_arg=False, want_arg_bytes=2)
    piece_row, first_piece = hr["piece_row"], hr["piece_ptr"][:-1]
    best = ops.segment_csr(out_h, hr["piece_ptr"], reduce)  # [H, K]: the rows' extremes (== out[rows])
    eq = out_h == best[piece_row]
    run = torch.cumsum(eq.to(torch.int32), 0)
    before = run[first_piece] - eq[first_piece].to(torch.int32)
    first = eq & ((run - before[piece_row]) == 1)
    return torch.where(first, bytes_h, torch.full_like(bytes_h, -1)).contiguous()


def spmm_planned(st: SparseStorage, weights: Optional[torch.Tensor], mat: torch.Tensor, reduce: str = "sum",
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reduce-SpMM of the matrix `st` describes (values `weights`, given separately; `out` only, no
    arg_out) with the storage's per-matrix choices: kernel family, COO row ids, compact copy of
    the hub rows.  Forward only (no autograd): the backward passes over the CSC view and the
    rank-local step of the multi-GPU product (distributed.py) are built on it.  `out`: fp32
    [M, K] to write into, possibly a column slice of a wider matrix."""
    algo = st._spmm_algo()
    col, row, hot_rows = st.col(), None, None
    lanes = 8 if mat.dtype in (torch.float16, torch.bfloat16) else 4  # elements per 16-byte lane of the edge-range kernels
    if algo == "edge_ranges" and mat.shape[1] % lanes == 0:
        row = st.row()
        plan = st._hot_columns()
        if plan is not None:
            hot_rows, col = ops._gather_rows_raw(mat, plan[0]), plan[1]
    else:
        algo = "auto"
    return ops._spmm(reduce, st.rowptr(), col, weights, mat, want_arg=False, row=row, algo=algo, hot_rows=hot_rows,
                     out=out, no_long_rows=st._longest_row() <= ops.LONG_ROW)[0]


def _spmm_sum_planned(st: SparseStorage, weights: Optional[torch.Tensor], mat: torch.Tensor) -> torch.Tensor:
    return spmm_planned(st, weights, mat, "sum")


def spmm_transposed_planned(st: SparseStorage, value: Optional[torch.Tensor], grad_out: torch.Tensor,
                            mean: bool = False) -> torch.Tensor:
    """A^T grad_out for the matrix `st` describes (values given separately; mean: weights over the
    row degree) — the gradient of spmm_sum / spmm_mean wrt the dense operand with a fixed adjacency:
    a planned forward over the CSC view with the weights brought to CSC order (memoised on `st`)."""
    csr2csc = st.csr2csc()  # leaves colptr and row[csr2csc] behind
    w = None
    if value is not None or mean:
        w = _csc_weights(st, value, csr2csc, st._row_in_csc_order(), mean)
    return spmm_planned(st._csc_view(), w, grad_out, "sum")


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value: Optional[torch.Tensor], mat: torch.Tensor,
                storage: SparseStorage, reduce: str, track: bool = True):
        rowptr, col = storage.rowptr(), storage.col()
        if mat.dtype in (torch.float16, torch.bfloat16):
            # half-width dense operand: 2-byte gathers and stores, fp32 sums (psa_spmm_half / psa_spmm_half_coo).
            # The backward stays half width too — the forward over the CSC view (fixed adjacency), one pass over
            # it for both gradients (trained values), the masked pass for min / max — and widens to fp32 only for
            # what those do not take: K % 8 != 0, K > 512, rows above 65 535 entries with min / max.
            need = track and (ctx.needs_input_grad[1] or (value is not None and ctx.needs_input_grad[0]))
            algo, row, hot_rows = "auto", None, None
            if mat.shape[1] % 8 == 0 and (value is None or value.dtype == torch.float32) and storage._spmm_algo() == "edge_ranges":
                # power-law matrix: the edge-range kernels and the hub-row copy, as for fp32
                algo, row = "edge_ranges", storage.row()
                plan = storage._hot_columns()
                if plan is not None:
                    hot_rows, col = ops._gather_rows_raw(mat.detach(), plan[0]), plan[1]
            arg_bytes = None
            need_mat = track and ctx.needs_input_grad[1]
            if (reduce in ("min", "max") and need and need_mat and _half_minmax_bw_ok(storage, mat.shape[1])
                    and (value is None or value.dtype in (torch.float32, mat.dtype))):
                # min / max whose backward will be the half-width masked pass over the CSC view: the forward leaves the
                # row-local form of arg_out only (1 byte per element up to 128-entry rows, 2 up to 65 535) — no int64
                # arg_out (8 bytes per element against the 2 of `out` itself); on a power-law matrix from the
                # edge-range kernels with the hub-row copy
                width = 2 if storage._longest_row() > ops.ARG_BYTES_EXACT_ROW else 1
                out, arg, arg_bytes = ops._spmm(reduce, rowptr, col, value, mat, want_arg=False, want_arg_bytes=width,
                                                row=row, algo=algo, hot_rows=hot_rows)
            else:
                out, arg = ops._spmm(reduce, rowptr, col, value, mat, want_arg=need and reduce in ("min", "max"), row=row,
                                     algo=algo, hot_rows=hot_rows)
            ctx.storage, ctx.reduce, ctx.half = storage, reduce, mat.dtype
            ctx.save_for_backward(value, mat, arg, arg_bytes, None)
            return out
        ctx.half = None
        algo = storage._spmm_algo()  # per-matrix choice, read once
        row = storage.row() if algo == "edge_ranges" else None  # the COO row ids the edge-balanced kernels walk
        hot_rows = None
        plan = storage._hot_columns() if algo == "edge_ranges" and mat.shape[1] % 4 == 0 else None
        if plan is not None:
            # hub columns: their rows of `mat` are gathered once into a compact copy (13 us for
            # 65 536 rows) and every reference goes there — consecutive addresses that stay cache
            # resident, instead of rows that may share a few memory channels
            hot_rows = ops.gather_rows(mat.detach(), plan[0])
            col = plan[1]
        arg = arg_bytes = piece_bytes = None
        if reduce in ("min", "max"):
            # What the backward will read decides what the forward stores.  The
            # int64 arg_out is two thirds of the forward's output traffic (2 GB of
            # 3.3 at 2 M x 128): it is skipped when nothing will be differentiated,
            # and when the one-pass backward over the CSC view will run (grad of
            # mat wanted, K tile supported) on a matrix whose row-local form of
            # arg_out is exact: one byte per element for rows of up to 128 entries,
            # two for rows of up to 65 536 (power-law graphs) — that form is then
            # the whole answer.
            need_value = track and value is not None and ctx.needs_input_grad[0]
            need_mat = track and ctx.needs_input_grad[1]
            csc_bw = need_mat and ops.minmax_bw_csc_supported(mat.shape[1])
            longest = storage._longest_row() if csc_bw else 0
            # rows above 65 535 entries (hubs of a large power-law graph): the two-byte form stays the whole answer
            # when those rows are cut into pieces that each keep an exact one (_huge_piece_winners)
            huge = (HUGE_ROW_PIECES and csc_bw and longest > ops.ARG_WORDS_EXACT_ROW
                    and storage._sparse_sizes[0] + 2 * HUGE_IDS < (1 << 31))
            width = 2 if (ops.ARG_BYTES_EXACT_ROW < longest <= ops.ARG_WORDS_EXACT_ROW or huge) else 1
            bytes_only = csc_bw and (longest <= ops.ARG_WORDS_EXACT_ROW or huge)
            want_arg = (need_value or need_mat) and not bytes_only
            res = ops._spmm(reduce, rowptr, col, value, mat, want_arg_bytes=width if csc_bw else False,
                            want_arg=want_arg, row=row, algo=algo, hot_rows=hot_rows,
                            no_long_rows=storage._longest_row() <= ops.LONG_ROW)
            out, arg, arg_bytes = res if csc_bw else (*res, None)
            if huge:
                piece_bytes = _huge_piece_winners(storage, reduce, value, mat)
        else:
            out = ops._spmm(reduce, rowptr, col, value, mat, row=row, algo=algo, hot_rows=hot_rows,
                            no_long_rows=storage._longest_row() <= ops.LONG_ROW)[0]
        ctx.storage, ctx.reduce = storage, reduce
        ctx.save_for_backward(value, mat, arg, arg_bytes, piece_bytes)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        value, mat, arg, arg_bytes, piece_bytes = ctx.saved_tensors
        st, reduce = ctx.storage, ctx.reduce
        need_value = value is not None and ctx.needs_input_grad[0]
        need_mat = ctx.needs_input_grad[1]
        grad_out = grad_out.contiguous()
        if (ctx.half is not None and reduce in ("sum", "mean") and need_mat and not need_value
                and grad_out.dtype == ctx.half and grad_out.shape[1] % 8 == 0):
            # fixed adjacency, half-width operands: A^T grad_out is the half-width forward over the
            # CSC view (fp32 weights, fp32 sums, one rounding) — no widening of grad_out, no fp32 pass
            csr2csc = st.csr2csc()
            mean = reduce == "mean"
            w = None
            if value is not None or mean:
                w = _csc_weights(st, None if value is None else value.detach().float(), csr2csc, st._row_in_csc_order(), mean)
            return None, _spmm_sum_planned(st._csc_view(), w, grad_out), None, None, None
        if (ctx.half is not None and reduce in ("sum", "mean") and need_value and need_mat and grad_out.dtype == ctx.half
                and ops.half_sum_bw_csc_supported(grad_out.shape[1])):
            # trained values, half-width operands: both gradients in one pass over the CSC view that gathers
            # 2-byte rows of grad_out and keeps fp32 sums — no fp32 copies of mat / grad_out (one wave per
            # column, columns above 128 entries — hub rows of a power-law matrix — in chunks of 128)
            csr2csc = st.csr2csc()
            v32 = value.detach().float()
            mean = reduce == "mean"
            back, forth = st._permute_plan("to_csr"), st._permute_plan("to_csc")
            folded = mean and back is not None and forth is not None
            # mean on the planned routes: 1 / deg(row) rides in the weights and is applied to grad_value after its
            # way back (SparseStorage._mean_scale_per_entry) instead of a dependent read per CSC entry inside the pass
            w = _streamed_values(st, v32 * st._mean_scale_per_entry() if folded else v32, plan=forth)
            if w is None:
                w = ops.transpose_weights(v32, csr2csc, None, None, False)
            scale = (1.0 / st.rowcount().clamp(min=1).to(torch.float32)) if mean and not folded else None
            gv, gm = ops.spmm_half_sum_bw_csc(st.colptr(), st._row_in_csc_order(), w, mat, grad_out, True, row_scale=scale,
                                              long_columns=st._csc_view()._longest_row() > ops.LONG_ROW)
            gv = ops.permute_apply(gv, back) if back is not None else ops.gather_rows(gv, st.csc2csr())
            if folded:
                gv = gv * st._mean_scale_per_entry()
            return gv.to(value.dtype), gm, None, None, None
        if (ctx.half is not None and reduce in ("min", "max") and arg_bytes is not None and need_mat
                and grad_out.dtype == ctx.half):
            # min / max, half-width operands: the masked half-width pass over the CSC view (both gradients when
            # the values are trained), fed by the row-local arg_out the forward left — no fp32 copies
            csr2csc = st.csr2csc()
            width = arg_bytes.element_size()
            w = None
            if value is not None:  # value[csr2csc]: the planned route when the storage has one (one request), else the gather
                w = _csc_weights(st, value.detach().float(), csr2csc, st._row_in_csc_order(), False)
            gv, gm = ops.spmm_half_minmax_bw_csc(st.colptr(), st._row_in_csc_order(), st._csc_edge_tags(width), w, mat,
                                                 grad_out, arg_bytes, want_value=need_value,
                                                 long_columns=st._csc_view()._longest_row() > ops.LONG_ROW)
            if gv is not None:
                plan = st._permute_plan("to_csr")
                gv = (ops.permute_apply(gv, plan) if plan is not None else ops.gather_rows(gv, st.csc2csr())).to(value.dtype)
            return gv, gm, None, None, None
        if ctx.half is not None:
            gv, gm = _SpMM._backward_fp32(st, reduce, None if value is None else value.float(), mat.float(),
                                          grad_out.float(), arg, None, need_value, need_mat)
            return (None if gv is None else gv.to(value.dtype), None if gm is None else gm.to(ctx.half),
                    None, None, None)
        gv, gm = _SpMM._backward_fp32(st, reduce, value, mat, grad_out, arg, arg_bytes, need_value, need_mat, piece_bytes)
        return gv, gm, None, None, None

    @staticmethod
    def _backward_fp32(st, reduce, value, mat, grad_out, arg, arg_bytes, need_value, need_mat, piece_bytes=None):
        grad_value = grad_mat = None
        if reduce in ("min", "max"):
            # With grad_mat wanted and a K tile the kernel takes, both gradients
            # come from ONE gather pass over the CSC view (no atomics): 256 M
            # scattered float atomics cost more than the one-off CSC build, which
            # the storage keeps for the next step.
            if need_mat and ops.minmax_bw_csc_supported(grad_out.shape[1]):
                csr2csc = st.csr2csc()  # first: it leaves colptr and row[csr2csc] behind
                width = arg_bytes.element_size() if arg_bytes is not None else 1
                view = st._csc_view()
                if piece_bytes is not None:
                    # rows above 65 535 entries: their CSC entries go to the pieces' rows of the compact copies
                    # (behind the view's own hub rows), tagged with their position inside the piece
                    ids_ext, row_ext, tags_ext, _ = st._huge_backward_plan()
                    if not need_value and view._spmm_algo() == "edge_ranges":
                        w = None if value is None else _csc_weights(st, value, csr2csc, st._row_in_csc_order(), False)
                        return None, ops.spmm_minmax_bw_eb(st.colptr(), view.row(), row_ext, tags_ext, w, grad_out, arg_bytes,
                                                           hot_ids=ids_ext, hot_bytes_tail=piece_bytes)
                    return ops.spmm_minmax_bw_csc(
                        st.rowptr(), st.colptr(), row_ext, csr2csc, tags_ext, value, mat, grad_out, None,
                        want_value=need_value, csc2csr=st.csc2csr() if need_value else None, arg_bytes=arg_bytes,
                        hot_ids=ids_ext, to_csr_plan=st._permute_plan("to_csr") if need_value else None,
                        value_csc=_streamed_values(st, value), hot_bytes_tail=piece_bytes)
                if not need_value and arg is None and arg_bytes is not None and view._spmm_algo() == "edge_ranges":
                    # fixed adjacency on a power-law matrix: the edge-range kernels over the CSC view, masked by
                    # the row-local arg_out (two gathers per entry), hub rows from compact copies
                    plan = view._hot_columns()
                    w = None if value is None else _csc_weights(st, value, csr2csc, st._row_in_csc_order(), False)
                    grad_mat = ops.spmm_minmax_bw_eb(st.colptr(), view.row(), st._row_in_csc_order() if plan is None else plan[1],
                                                     st._csc_edge_tags(width), w, grad_out, arg_bytes,
                                                     hot_ids=None if plan is None else plan[0])
                    return None, grad_mat
                # hub rows of a power-law matrix: the pass reads their rows of grad_out and of
                # arg_bytes from compact copies (the CSC view's own hot "columns")
                # (with the two-byte form only: a matrix whose rows all fit the one-byte form has no hub rows to speak of)
                plan = st._csc_view()._hot_columns() if (arg is None and arg_bytes is not None and width == 2) else None
                grad_value, grad_mat = ops.spmm_minmax_bw_csc(
                    st.rowptr(), st.colptr(), st._row_in_csc_order() if plan is None else plan[1], csr2csc,
                    st._csc_edge_tags(width), value, mat, grad_out, arg, want_value=need_value,
                    csc2csr=st.csc2csr() if need_value else None, arg_bytes=arg_bytes,
                    hot_ids=None if plan is None else plan[0],
                    to_csr_plan=st._permute_plan("to_csr") if need_value else None,
                    value_csc=_streamed_values(st, value) if arg is None else None)
            else:
                grad_value, grad_mat = ops.spmm_minmax_bw(st.col(), value, mat, grad_out, arg,
                                                         want_value=need_value, want_mat=need_mat)
        else:
            mean = reduce == "mean"
            if need_value and need_mat and ops.minmax_bw_csc_supported(grad_out.shape[1]):
                # Trainable edge values: both gradients in ONE pass over the CSC
                # view — the gathered grad_out row feeds grad_mat and, dotted with
                # the column's own mat row, grad_value (instead of a second full
                # gather of mat rows in spmm_value_bw plus the weight gather).
                csr2csc = st.csr2csc()
                back, forth = st._permute_plan("to_csr"), st._permute_plan("to_csc")
                folded = mean and back is not None and forth is not None and value.dtype == torch.float32
                # mean: 1/deg(row) multiplies both gradients per edge — inside the pass (row_scale, a dependent read per
                # CSC entry), or, on the planned routes, folded into the weights before their way to CSC order and into
                # grad_value after its way back (two streaming multiplies; SparseStorage._mean_scale_per_entry)
                scale = (1.0 / st.rowcount().clamp(min=1).to(torch.float32)) if mean and not folded else None
                plan = st._csc_view()._hot_columns()  # hub rows: grad_out rows from a compact copy
                streamed = _streamed_values(st, value.detach() * st._mean_scale_per_entry() if folded else value, plan=forth)
                grad_value, grad_mat = ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order() if plan is None else plan[1],
                                                           csr2csc, value, mat, grad_out, True, csc2csr=st.csc2csr(),
                                                           row_scale=scale, hot_ids=None if plan is None else plan[0],
                                                           to_csr_plan=back, value_csc=streamed)
                if folded:
                    grad_value = grad_value * st._mean_scale_per_entry()
                return grad_value, grad_mat
            if need_value:
                grad_value = ops.spmm_value_bw(None, st.rowptr(), st.col(), mat, grad_out,
                                               "mean" if mean else "sum")
            if need_mat:
                # A^T grad_out = a forward SpMM over the CSC view, with that view's own choices
                # (edge ranges and the hub-row copy when the transpose is a power-law matrix too)
                grad_mat = spmm_transposed_planned(st, value, grad_out, mean)
        return grad_value, grad_mat


def spmm_sparse(src: SparseTensor, other: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """out = reduce-SpMM(src, other) for a dense `other` [N, K] (fp32)."""
    if reduce == "add":
        reduce = "sum"
    if reduce not in ("sum", "mean", "min", "max"):
        raise ValueError(reduce)
    if other.dim() != 2 or other.shape[0] != src.sparse_size(1):
        raise ValueError(f"dense operand must be [{src.sparse_size(1)}, K]")
    value = src.storage.value()
    if value is not None and value.dim() != 1:
        raise ValueError("spmm needs scalar edge values")
    # forward() runs with grad mode off and sees requires_grad flags only: whether a
    # backward can follow at all is decided here
    return _SpMM.apply(value, other, src.storage, reduce, torch.is_grad_enabled())


def matmul(src: SparseTensor, other, reduce: str = "sum") -> torch.Tensor:
    if isinstance(other, torch.Tensor):
        return spmm_sparse(src, other, reduce)
    if isinstance(other, SparseTensor):
        if reduce not in ("sum", "add"):
            raise NotImplementedError("sparse @ sparse supports reduce='sum' only")
        from .spspmm import spspmm_tensor

        return spspmm_tensor(src, other)
    raise ValueError("matmul: `other` must be a dense torch.Tensor or a SparseTensor")


def spmm(index: torch.Tensor, value: Optional[torch.Tensor], m: int, n: int,
         matrix: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """README.md:269-285 functional form: (index, value) COO of an m x n matrix
    times `matrix` [n, K].  Entries are sorted (duplicates added) first."""
    index, value = coalesce(index, value, m, n, op="add")
    src = SparseTensor(row=index[0].contiguous(), col=index[1].contiguous(), value=value,
                       sparse_sizes=(m, n), is_sorted=True, trust_data=True)
    return spmm_sparse(src, matrix, reduce)


SparseTensor.spmm = lambda self, other, reduce="sum": spmm_sparse(self, other, reduce)
SparseTensor.matmul = lambda self, other, reduce="sum": matmul(self, other, reduce)
SparseTensor.__matmul__ = lambda self, other: matmul(self, other, "sum")
