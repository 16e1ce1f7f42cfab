"""Operator surface of the HIP core — the counterpart of the reference's
`paddle_sparse_ops` extension module (csrc/convert.cpp:37-43,70-76,
csrc/version.cpp:35-40), same names and argument meaning, on torch tensors
that live in MI355X HBM.

torch is plumbing here (device memory + the current HIP stream); every op is
one or more calls through the C-ABI in include/paddle_sparse_hip.h.  Outputs
are allocated by the caller side (here: torch's allocator), as the reference
does with paddle::empty.  Calls are asynchronous on the current stream.

CPU tensors are rejected: this build has no CPU kernels and never falls back.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import REDUCE_ID, HipCoreError, check

_check = check  # for functions with a parameter named `check`


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """The current HIP stream of the current device as an integer handle (2.9 us through
    torch.cuda.current_stream(), which builds a Stream object; 0.3 us through the raw getter)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Context that makes `device` current for the launches inside — the no-op
    singleton when it already is (torch.cuda.device() costs ~8 us per op even
    then, as much as two kernel launches)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _gpu(x: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(
            f"{name} must be a GPU tensor: paddle_sparse_amd runs on MI355X "
            "only and has no CPU path")
    return x


def _index(x: torch.Tensor, name: str) -> torch.Tensor:
    _gpu(x, name)
    if x.dtype != torch.int64:
        raise TypeError(f"{name} must be int64 (got {x.dtype})")
    if x.dim() != 1:
        raise ValueError(f"{name} must be 1-D")
    return x.contiguous()


def _ptr(x: Optional[torch.Tensor]) -> Optional[int]:
    return None if x is None else x.data_ptr()


def sparse_cuda_version() -> torch.Tensor:
    """csrc/version.cpp:14-22: int64[1] on the CPU; -1 on this (HIP) build."""
    return torch.full((1,), _lib.load().psa_sparse_cuda_version(), dtype=torch.int64)


def ind2ptr(ind: torch.Tensor, M: int) -> torch.Tensor:
    """csrc/convert.cpp:13-43.  ind: sorted int64[E] -> int64[M+1]."""
    ind = _index(ind, "ind")
    out = torch.empty(M + 1, dtype=torch.int64, device=ind.device)
    with _on(ind.device):
        check(_lib.load().psa_ind2ptr(_ptr(ind), ind.numel(), M, _ptr(out), _stream()))
    return out


def ptr2ind(ptr: torch.Tensor, E: int) -> torch.Tensor:
    """csrc/convert.cpp:46-76.  ptr: int64[M+1] -> int64[E]."""
    ptr = _index(ptr, "ptr")
    if ptr.numel() < 1:
        raise ValueError("ptr must have at least one element")
    out = torch.empty(E, dtype=torch.int64, device=ptr.device)
    with _on(ptr.device):
        check(_lib.load().psa_ptr2ind(_ptr(ptr), ptr.numel() - 1, E, _ptr(out), _stream()))
    return out


def _spmm(reduce: str, rowptr: torch.Tensor, col: torch.Tensor,
          value: Optional[torch.Tensor], mat: torch.Tensor, want_arg_bytes: bool = False,
          want_arg: bool = True, row: Optional[torch.Tensor] = None, algo: str = "auto",
          out: Optional[torch.Tensor] = None, hot_rows: Optional[torch.Tensor] = None,
          no_long_rows: bool = False):
    """(out, arg_out | None) — and, with want_arg_bytes (min/max, K % 4 == 0), a
    third result: arg_out as row-local byte indices for spmm_minmax_bw_csc.
    want_arg=False (min/max) skips the int64 arg_out altogether: the kernel then
    stores `out` (and the bytes, if asked for) only — two thirds of the output
    traffic gone; for callers that need no backward, or whose backward is served
    by the bytes alone (no row longer than ARG_BYTES_EXACT_ROW entries).
    `row` (the COO row ids, SparseStorage.row()) is optional: the edge-balanced
    kernels read it and derive it from rowptr when it is not given.  `algo`:
    "auto" | "row_waves" | "edge_ranges" (psa_spmm_algo; csr_row_stats /
    SparseStorage._spmm_algo() choose per matrix).  `out`: write into this
    fp32 [M, K] tensor, which may be a column slice of a wider contiguous
    matrix (row stride = the wider matrix's width).  `hot_rows` (edge_ranges
    only): float32 [h, K] compact copy of rows of mat; column ids in [N, N + h)
    name its rows (SparseStorage._hot_columns builds the redirected col).  `no_long_rows`: the
    caller knows that no row has more than 128 entries (SparseStorage._longest_row); for K <= 128 the
    call then brings no long-row workspace, which saves the three near-empty launches of the long-row
    machinery — 5 us of a 45 us problem such as BASELINE config 2."""
    rowptr = _index(rowptr, "rowptr")
    col = _index(col, "col")
    if row is not None:
        row = _index(row, "row")
        if row.numel() != col.numel():
            raise ValueError("row must have one entry per edge")
    _gpu(mat, "mat")
    if mat.dtype in (torch.float16, torch.bfloat16):
        if out is not None:
            raise ValueError("the half-width SpMM does not take `out`")
        res = _spmm_half(reduce, rowptr, col, value, mat, want_arg, row=row, algo=algo, hot_rows=hot_rows,
                         want_arg_bytes=want_arg_bytes)
        return res
    if mat.dtype != torch.float32:
        raise TypeError(f"spmm takes float32, float16 or bfloat16 dense operands (mat is {mat.dtype})")
    if mat.dim() != 2:
        raise ValueError("mat must be 2-D [N, K]")
    mat = mat.contiguous()
    if value is not None:
        _gpu(value, "value")
        if value.dtype != torch.float32 or value.dim() != 1 or value.numel() != col.numel():
            raise ValueError("value must be float32[nnz]")
        value = value.contiguous()
    if rowptr.numel() < 1:
        raise ValueError("rowptr must have at least one element")
    M, (N, K), nnz = rowptr.numel() - 1, mat.shape, col.numel()
    rid = REDUCE_ID[reduce]
    num_hot = 0
    if hot_rows is not None:
        _gpu(hot_rows, "hot_rows")
        if hot_rows.dtype != torch.float32 or hot_rows.dim() != 2 or hot_rows.shape[1] != K or not hot_rows.is_contiguous():
            raise ValueError("hot_rows must be a contiguous float32 [h, K] tensor")
        num_hot = hot_rows.shape[0]
    ldo = 0
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=mat.device)
    else:
        _gpu(out, "out")
        if out.dtype != torch.float32 or out.shape != (M, K) or (K > 1 and out.stride(1) != 1):
            raise ValueError("out must be float32 [M, K] with unit column stride")
        ldo = out.stride(0) if M > 1 else K
        if ldo < K:
            raise ValueError("out rows overlap")
    arg = None
    minmax = rid in (_lib.MIN, _lib.MAX)
    lib = _lib.load()
    ws_bytes = lib.psa_spmm_workspace_bytes(rid, K, nnz)  # long-row scratch (0 if no row can be long)
    if no_long_rows and algo != "edge_ranges" and (K <= 64 or (K <= 128 and not want_arg_bytes and N * K * 4 < (6 << 30))):
        # every row on its own wave / lane group: same results, no list / chunk / combine launches (K <= 64: the
        # multirow kernel; 64 < K <= 128: the row kernel, as fast as the fused-roles one on rows this short —
        # 100k x 100k, F = 128: 83.5 -> 76.0 us, config 3: 1608 -> 1603 us; from K = 192 the fused kernel's
        # 128-wide tiles win, and only it gathers non-temporally for operands beyond 6 GiB)
        ws_bytes = 0
    ws = _workspace(ws_bytes, mat.device) if ws_bytes else None
    # the kernels that write the byte form themselves: K <= 64 (multirow) and, for
    # 64 < K <= 256, the fused-roles kernel — which spmm_dispatch takes only when a row
    # CAN be long (nnz > kLongRow = 128; the edge-range kernels write the bytes too, but
    # fall back to the row path on shapes they do not take).  Behind the others the
    # bytes are derived from arg_out, which must then exist.  (ws_bytes > 0 is not the
    # test: it also covers the edge-range scratch, which every nnz > 0 has.)
    bytes_in_kernel = K % 4 == 0 and (K <= 64 or (K <= 256 and nnz > LONG_ROW))
    if minmax and want_arg_bytes and K % 4 == 0 and K <= 256 and not bytes_in_kernel:
        want_arg = True
    if minmax and want_arg:
        arg = torch.empty((M, K), dtype=torch.int64, device=mat.device)
    arg_bytes = None
    arg_width = 2 if want_arg_bytes == 2 else 1  # True / 1: one byte per element; 2: two (exact up to 65 536-entry rows)
    if want_arg_bytes and minmax and K % 4 == 0 and (want_arg or K <= 256):
        arg_bytes = torch.empty((M, K), dtype=torch.int16 if arg_width == 2 else torch.uint8, device=mat.device)
    with _on(mat.device):
        check(lib.psa_spmm_coo(rid, _ptr(rowptr), _ptr(row), _ptr(col), _ptr(value), _ptr(mat),
                               _ptr(hot_rows) if num_hot else None, num_hot, M, N, K, nnz, _ptr(out), ldo, _ptr(arg), _ptr(arg_bytes),
                               arg_width, _lib.SPMM_ALGO_ID[algo], _ptr(ws), ws_bytes, _stream()))
    if want_arg_bytes:
        return out, arg, arg_bytes
    return out, arg


def _spmm_half(reduce: str, rowptr, col, value, mat, want_arg: bool = True, row=None, algo: str = "auto",
               hot_rows=None, want_arg_bytes=False):
    """fp16 / bf16 `mat` -> (out in mat's dtype, arg_out | None): fp32 products and sums,
    one rounding on store (psa_spmm_half / psa_spmm_half_coo).  value: None, float32[nnz] or
    mat's dtype[nnz].  algo="edge_ranges" (with `row`, `hot_rows` as for fp32; fp32 or no values):
    the edge-balanced kernels.  K % 8 != 0 widens to fp32 and rounds the fp32 kernel's result."""
    if mat.dim() != 2:
        raise ValueError("mat must be 2-D [N, K]")
    mat = mat.contiguous()
    M, (N, K), nnz = rowptr.numel() - 1, mat.shape, col.numel()
    rid = REDUCE_ID[reduce]
    minmax = rid in (_lib.MIN, _lib.MAX)
    if value is not None:
        _gpu(value, "value")
        if value.dtype not in (torch.float32, mat.dtype) or value.dim() != 1 or value.numel() != nnz:
            raise ValueError("value must be float32[nnz] or mat's dtype[nnz]")
        value = value.contiguous()
    if want_arg_bytes and K % 8 != 0:
        raise ValueError("the half-width SpMM writes arg_bytes for K % 8 == 0 only")
    if K % 8 != 0:
        res = _spmm(reduce, rowptr, col, None if value is None else value.float(), mat.float(), want_arg=want_arg)
        return res[0].to(mat.dtype), res[1]
    out = torch.empty((M, K), dtype=mat.dtype, device=mat.device)
    arg = torch.empty((M, K), dtype=torch.int64, device=mat.device) if minmax and want_arg else None
    if algo == "edge_ranges" and nnz > 0 and (value is None or value.dtype == torch.float32):
        num_hot = 0
        if hot_rows is not None:
            _gpu(hot_rows, "hot_rows")
            if hot_rows.dtype != mat.dtype or hot_rows.dim() != 2 or hot_rows.shape[1] != K or not hot_rows.is_contiguous():
                raise ValueError("hot_rows must be a contiguous [h, K] tensor of mat's dtype")
            num_hot = hot_rows.shape[0]
        lib = _lib.load()
        ws = _workspace(lib.psa_spmm_half_workspace_bytes(rid, K, nnz), mat.device)
        eb_bytes, eb_width = None, 2 if want_arg_bytes == 2 else 1
        if want_arg_bytes and minmax:  # the row-local form, from the edge-range kernels (power-law matrices)
            eb_bytes = torch.empty((M, K), dtype=torch.int16 if eb_width == 2 else torch.uint8, device=mat.device)
        with _on(mat.device):
            check(lib.psa_spmm_half_coo(rid, _DTYPE_ID[mat.dtype], _ptr(rowptr), _ptr(row), _ptr(col), _ptr(value), _ptr(mat),
                                        _ptr(hot_rows) if num_hot else None, num_hot, M, N, K, nnz, _ptr(out), _ptr(arg),
                                        _ptr(eb_bytes), eb_width, _lib.SPMM_ALGO_ID[algo], _ptr(ws), ws.numel(), _stream()))
        if want_arg_bytes:
            return out, arg, eb_bytes
        return out, arg
    if hot_rows is not None:
        raise ValueError("hot_rows needs algo='edge_ranges' and fp32 (or no) values")
    arg_bytes, width = None, 2 if want_arg_bytes == 2 else 1
    if want_arg_bytes and minmax:  # the row-local form for the half-width one-pass backward (one wave per row only)
        arg_bytes = torch.empty((M, K), dtype=torch.int16 if width == 2 else torch.uint8, device=mat.device)
    with _on(mat.device):
        check(_lib.load().psa_spmm_half_arg(rid, _DTYPE_ID[mat.dtype], _ptr(rowptr), _ptr(col), _ptr(value),
                                            _DTYPE_ID[value.dtype] if value is not None else 0, _ptr(mat), M, N, K, nnz,
                                            _ptr(out), _ptr(arg), _ptr(arg_bytes), width, _stream()))
    if want_arg_bytes:
        return out, arg, arg_bytes
    return out, arg


# rows above this many entries leave their wave for the chunk role (psa::kLongRow, csrc/long_rows.h)
LONG_ROW = 128
# rows up to this many entries: the one-byte form of arg_out (arg_bytes) needs no arg_out beside it
ARG_BYTES_EXACT_ROW = 128
# ... and the two-byte form (want_arg_bytes=2)
ARG_WORDS_EXACT_ROW = 65_535  # 0xffff is "no winner" (vec_io.h)


def spmm_sum(rowptr, col, value, mat, row=None, algo="auto") -> torch.Tensor:
    """out[i] = sum_e value[e] * mat[col[e]] (value None -> weights 1)."""
    return _spmm("sum", rowptr, col, value, mat, row=row, algo=algo)[0]


def spmm_mean(rowptr, col, value, mat, row=None, algo="auto") -> torch.Tensor:
    return _spmm("mean", rowptr, col, value, mat, row=row, algo=algo)[0]


def spmm_min(rowptr, col, value, mat, row=None, algo="auto") -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (out, arg_out); arg_out == nnz marks an empty row."""
    return _spmm("min", rowptr, col, value, mat, row=row, algo=algo)


def spmm_max(rowptr, col, value, mat, row=None, algo="auto") -> Tuple[torch.Tensor, torch.Tensor]:
    return _spmm("max", rowptr, col, value, mat, row=row, algo=algo)


def csr_row_stats(rowptr: torch.Tensor) -> Tuple[int, int, int, int]:
    """(rows without entries, rows with 1-2 entries, rows above 128 entries,
    longest row) of a CSR pointer.  Synchronises (one 32-byte host read)."""
    rowptr = _index(rowptr, "rowptr")
    stats = torch.empty(4, dtype=torch.int64, device=rowptr.device)
    with _on(rowptr.device):
        check(_lib.load().psa_csr_row_stats(_ptr(rowptr), rowptr.numel() - 1, _ptr(stats), _stream()))
    e, t, b, m = stats.tolist()
    return e, t, b, m


def spmm_set_variant(variant: int) -> int:
    """Bench/test hook: pick the SpMM kernel variant (0 = auto)."""
    return _lib.load().psa_spmm_set_variant(int(variant))


# ---------------------------------------------------------------------------
# sort / gather / coalesce building blocks
# ---------------------------------------------------------------------------

_DTYPE_ID = {
    torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3,
    torch.float16: 4, torch.bfloat16: 5,
}


_POISON_WORKSPACE = os.environ.get("PSA_POISON_WORKSPACE") == "1"  # test hook: scratch starts as 0xff, not as whatever it held


def _workspace(nbytes: int, device) -> torch.Tensor:
    # torch's caching allocator hands back >= 512-byte aligned blocks
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    if _POISON_WORKSPACE:
        ws.fill_(255)
    return ws


class SortScratch:
    """Workspace of a finished sort, kept alive so that the next host read can fold in the
    sort's look-back diagnostic (unique_sorted(after=...))."""

    def __init__(self, ws: torch.Tensor, n: int, max_value: int):
        self.ws, self.n, self.max_value = ws, n, max_value


def _raise_on_sort_fault(lib, ws: torch.Tensor, n: int, max_value: int, who: str) -> None:
    """One 4-byte host read behind a sort that no other host read follows: the look-back
    diagnostic of the single-sweep passes (psa_index_sort_status).  A wait that gave up has
    made the last pass store -1 instead of an order; here it becomes an exception, as the
    reference's argsort could never hand back a wrong permutation (storage.py:164-169)."""
    status = lib.psa_index_sort_status(_ptr(ws), n, max_value, _stream())
    if status != 0:
        raise HipCoreError(f"{who}: an inter-workgroup wait of a radix pass gave up "
                           f"(status {status}); the order is invalid")


def index_sort(keys: torch.Tensor, max_value: Optional[int] = None,
               with_sorted_inputs: bool = False, keep_scratch: bool = False, check: bool = False):
    """paddle_sparse/utils.py:14-23 — returns (sorted | None, perm).

    Stable LSD radix sort in HIP; `max_value` (exclusive bound on the keys, the
    reference passes M*N) selects the number of 8-bit passes.  perm is the
    stable sorting permutation, bit-identical to numpy argsort(kind="stable").
    keep_scratch: also return a SortScratch for unique_sorted(after=...).
    check: read the sort's fault word afterwards (one host read, synchronises) and
    raise HipCoreError if a look-back wait gave up — for callers whose result no other
    host read follows (SparseStorage's constructor sort, csr2csc).
    ONLY THE FAULT WORD IS AUTHORITATIVE: after a fault the outputs hold -1 where the
    faulted tiles of the last pass sat, possibly valid-looking entries and untouched memory
    elsewhere — the absence of -1 in a part of them proves nothing.  A caller that passes
    check=False must read the word itself before it trusts the order: keep_scratch=True and
    unique_sorted(after=...) (the coalesce path: the word comes back with the count, no
    extra synchronisation) or index_sort_checked / sort_fault_word.
    """
    keys = _index(keys, "keys")
    n = keys.numel()
    if max_value is None:
        max_value = (int(keys.max()) + 1) if n else 1
    max_value = max(int(max_value), 1)
    perm = torch.empty(n, dtype=torch.int64, device=keys.device)
    out = torch.empty(n, dtype=torch.int64, device=keys.device) if with_sorted_inputs else None
    lib = _lib.load()
    nbytes = lib.psa_index_sort_workspace_bytes(n, max_value)
    ws = _workspace(nbytes, keys.device)
    with _on(keys.device):
        _check(lib.psa_index_sort(_ptr(keys), n, max_value, _ptr(out), _ptr(perm),
                                  _ptr(ws), ws.numel(), _stream()))
        if check and n > 0:
            _raise_on_sort_fault(lib, ws, n, max_value, "index_sort")
    if keep_scratch:
        return out, perm, SortScratch(ws, n, max_value)
    return out, perm


def index_sort_checked(keys: torch.Tensor, max_value: int):
    """index_sort plus the look-back diagnostic word (tests): returns
    (sorted, perm, status) with status 0 when every inter-workgroup wait of the
    single-sweep passes ended normally.  Synchronises."""
    keys = _index(keys, "keys")
    n = keys.numel()
    max_value = max(int(max_value), 1)
    perm = torch.empty(n, dtype=torch.int64, device=keys.device)
    out = torch.empty(n, dtype=torch.int64, device=keys.device)
    lib = _lib.load()
    ws = _workspace(lib.psa_index_sort_workspace_bytes(n, max_value), keys.device)
    with _on(keys.device):
        check(lib.psa_index_sort(_ptr(keys), n, max_value, _ptr(out), _ptr(perm),
                                 _ptr(ws), ws.numel(), _stream()))
        status = lib.psa_index_sort_status(_ptr(ws), n, max_value, _stream())
    return out, perm, status


def sort_pairs(keys: torch.Tensor, payload: torch.Tensor, max_value: Optional[int] = None, keep_scratch: bool = False,
               check: bool = False):
    """Stable sort of (key, 4-byte payload) pairs: returns (sorted_keys,
    payload[perm]) without ever forming perm.  payload: 1-D, 4-byte dtype.
    keep_scratch: also return a SortScratch for unique_sorted(after=...).
    check: as for index_sort."""
    keys = _index(keys, "keys")
    _gpu(payload, "payload")
    if payload.dim() != 1 or payload.element_size() != 4 or payload.numel() != keys.numel():
        raise ValueError("payload must be 1-D, 4 bytes per element, same length as keys")
    payload = payload.contiguous()
    n = keys.numel()
    if max_value is None:
        max_value = (int(keys.max()) + 1) if n else 1
    max_value = max(int(max_value), 1)
    out_keys = torch.empty_like(keys)
    out_pay = torch.empty_like(payload)
    lib = _lib.load()
    ws = _workspace(lib.psa_index_sort_workspace_bytes(n, max_value), keys.device)
    with _on(keys.device):
        _check(lib.psa_sort_pairs_u32(_ptr(keys), _ptr(payload), n, max_value, _ptr(out_keys),
                                      _ptr(out_pay), _ptr(ws), ws.numel(), _stream()))
        if check and n > 0:
            _raise_on_sort_fault(lib, ws, n, max_value, "sort_pairs")
    if keep_scratch:
        return out_keys, out_pay, SortScratch(ws, n, max_value)
    return out_keys, out_pay


def sort_pairs_field(keys: torch.Tensor, payload: torch.Tensor, first_bit: int, max_value: int,
                     keep_scratch: bool = False):
    """Stable sort of (key, 4-byte payload) pairs by the bit field
    (key >> first_bit) < max_value only; the low bits ride inside the key.
    keep_scratch: also return a SortScratch for unique_sorted(after=...)."""
    keys = _index(keys, "keys")
    _gpu(payload, "payload")
    if payload.dim() != 1 or payload.element_size() != 4 or payload.numel() != keys.numel():
        raise ValueError("payload must be 1-D, 4 bytes per element, same length as keys")
    payload = payload.contiguous()
    n, max_value = keys.numel(), max(int(max_value), 1)
    out_keys, out_pay = torch.empty_like(keys), torch.empty_like(payload)
    lib = _lib.load()
    ws = _workspace(lib.psa_index_sort_workspace_bytes(n, max_value), keys.device)
    with _on(keys.device):
        check(lib.psa_sort_pairs_u32_field(_ptr(keys), _ptr(payload), n, int(first_bit), max_value,
                                           _ptr(out_keys), _ptr(out_pay), _ptr(ws), ws.numel(), _stream()))
    if keep_scratch:
        return out_keys, out_pay, SortScratch(ws, n, max_value)
    return out_keys, out_pay


def merge_sorted(a: torch.Tensor, b: torch.Tensor, payload_a: Optional[torch.Tensor] = None,
                 payload_b: Optional[torch.Tensor] = None, want_source: bool = True):
    """Stable merge of two SORTED int64 key arrays (a's entry first on ties):
    what index_sort(cat([a, b])) returns, without the sort.  Returns (merged,
    source | None, payload | None); source indexes the concatenation, payload
    (when both 4-byte 1-D payload arrays are given) is cat([pa, pb])[source]."""
    a, b = _index(a, "a"), _index(b, "b")
    na, nb = a.numel(), b.numel()
    pay_out = None
    if payload_a is not None or payload_b is not None:
        for name, p, n in (("payload_a", payload_a, na), ("payload_b", payload_b, nb)):
            if p is None:
                raise ValueError("merge_sorted: give both payloads or neither")
            _gpu(p, name)
            if p.dim() != 1 or p.element_size() != 4 or p.numel() != n or not p.is_contiguous():
                raise ValueError(f"{name} must be contiguous, 1-D, 4 bytes per element, one per key")
        if payload_a.dtype != payload_b.dtype:
            raise ValueError("merge_sorted: payload dtypes differ")
        pay_out = torch.empty(na + nb, dtype=payload_a.dtype, device=a.device)
    merged = torch.empty(na + nb, dtype=torch.int64, device=a.device)
    source = torch.empty(na + nb, dtype=torch.int64, device=a.device) if want_source else None
    with _on(a.device):
        check(_lib.load().psa_merge_sorted(_ptr(a), na, _ptr(b), nb, _ptr(payload_a), _ptr(payload_b),
                                           _ptr(merged), _ptr(source), _ptr(pay_out), _stream()))
    return merged, source, pay_out


def coalesce_small_max() -> int:
    return int(_lib.load().psa_coalesce_small_max())


def coalesce_small(row: torch.Tensor, col: torch.Tensor, m: int, n: int):
    """Sort by (row, col) + run-length structure in ONE launch of one workgroup
    (inputs of at most coalesce_small_max() entries; see the header).  Returns
    (count, ptr int64[count + 1], row' int64[count], col' int64[count], perm
    int64[nnz]) like index_sort + unique_sorted.  One host read (count)."""
    row, col = _index(row, "row"), _index(col, "col")
    nnz = row.numel()
    dev = row.device
    out = torch.empty((2, nnz), dtype=torch.int64, device=dev)
    ptr = torch.empty(nnz + 1, dtype=torch.int64, device=dev)
    perm = torch.empty(nnz, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    lib = _lib.load()
    ws = _workspace(lib.psa_coalesce_small_workspace_bytes(nnz), dev)
    with _on(dev):
        check(lib.psa_coalesce_small(_ptr(row), _ptr(col), nnz, int(m), int(n), out[0].data_ptr(),
                                     out[1].data_ptr(), _ptr(ptr), _ptr(perm), _ptr(count), _ptr(ws),
                                     ws.numel(), _stream()))
    c = int(count.item())
    return c, ptr[:c + 1], out[0, :c], out[1, :c], perm


class IndexRangeError(AssertionError):
    """A row / col index lies outside the matrix (the reference asserts
    row.max() < M and col.max() < N, storage.py:78-91)."""


_FUSED_SMALL = 10_240  # entries the one-launch coalesce takes (psa_coalesce_small_max_fused)


def _check_chain_flags(flags: int, m: int, n: int) -> None:
    if flags & 1:
        raise IndexRangeError(f"coalesce: an index lies outside the {m} x {n} matrix")
    if flags & 4:
        raise HipCoreError("coalesce: an inter-workgroup wait of the radix sort gave up; the result is invalid")


def coalesce_chain(row: torch.Tensor, col: torch.Tensor, value: Optional[torch.Tensor], m: int, n: int,
                   op: str = "sum", read_first: bool = False):
    """coalesce of (row, col, value) of an m x n matrix in two C-ABI calls
    (psa_coalesce_count / psa_coalesce_write) and ONE host read.  Returns
    (index int64[2, count] contiguous, value' | None, was_sorted).  With
    read_first the count is read between the calls and the outputs are sized
    exactly; otherwise both calls are enqueued on worst-case buffers first."""
    row, col = _index(row, "row"), _index(col, "col")
    nnz, dev = row.numel(), row.device
    if col.numel() != nnz:
        raise ValueError("row and col must have the same length")
    dt, D = 0, 0
    if value is not None:
        _gpu(value, "value")
        if value.dtype not in _DTYPE_ID:
            raise TypeError(f"coalesce: unsupported value dtype {value.dtype}")
        if value.shape[0] != nnz:
            raise ValueError("value.shape[0] must equal nnz")
        value = value.contiguous()
        dt, D = _DTYPE_ID[value.dtype], 1
        for s in value.shape[1:]:
            D *= s
    if nnz == 0:
        return torch.empty((2, 0), dtype=torch.int64, device=dev), value, True
    lib = _lib.load()
    rid = REDUCE_ID[op]
    if (nnz <= _FUSED_SMALL and not read_first and 0 < m * n < (1 << 62)
            and (value is None or (D == 1 and value.dim() == 1 and value.dtype in (torch.float32, torch.int32)))):
        # one launch for everything (one workgroup, sort resident in the LDS): one buffer holds the
        # index (2 nnz words), the values (nnz 4-byte words) and the two status words
        buf = torch.empty(3 * nnz + 2, dtype=torch.int64, device=dev)
        status = buf[3 * nnz:]
        out = buf[2 * nnz:3 * nnz].view(value.dtype)[:nnz] if value is not None else None
        with _on(dev):
            check(lib.psa_coalesce_small_fused(_ptr(row), _ptr(col), _ptr(value), dt, nnz, int(m), int(n), rid,
                                               _ptr(buf), _ptr(out), _ptr(status), _stream()))
        count, flags = status.tolist()
        _check_chain_flags(flags, m, n)
        return buf[:2 * count].view(2, count), (None if out is None else out[:count]), not (flags & 2)
    ws = _workspace(lib.psa_coalesce_workspace_bytes(nnz, int(m), int(n)), dev)
    status = ws[:16].view(torch.int64)
    tail = tuple(value.shape[1:]) if value is not None else ()
    with _on(dev):
        check(lib.psa_coalesce_count(_ptr(row), _ptr(col), _ptr(value), dt, D, nnz, int(m), int(n),
                                     _ptr(ws), ws.numel(), _stream()))
        count = -1
        if read_first:
            count, flags = status.tolist()
            _check_chain_flags(flags, m, n)
        rows = nnz if count < 0 else count
        index = torch.empty(2 * rows, dtype=torch.int64, device=dev)
        out = torch.empty((rows,) + tail, dtype=value.dtype, device=dev) if value is not None else None
        check(lib.psa_coalesce_write(_ptr(value), dt, D, nnz, int(m), int(n), rid, count, _ptr(ws),
                                     _ptr(index), _ptr(out), _stream()))
        if not read_first:
            count, flags = status.tolist()
            _check_chain_flags(flags, m, n)
    index = index[:2 * count].view(2, count)
    if out is not None:
        out = out[:count]
    return index, out, not (flags & 2)


def make_keys_checked(row: torch.Tensor, col: torch.Tensor, m: int, n: int):
    """keys = row * n + col plus the device status words {0, flags}: bit 0 = an
    index outside [0, m) x [0, n), bit 1 = keys not sorted (storage.py:159-163)."""
    row, col = _index(row, "row"), _index(col, "col")
    keys = torch.empty_like(row)
    status = torch.empty(4, dtype=torch.int64, device=row.device)
    with _on(row.device):
        check(_lib.load().psa_make_keys_checked(_ptr(row), _ptr(col), row.numel(), int(m), int(n), _ptr(keys),
                                                _ptr(status), _stream()))
    return keys, status


def make_keys(a: torch.Tensor, b: torch.Tensor, mul: int, check_sorted: bool = False
              ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """keys = a * mul + b  (+ device flag "some key is smaller than its
    predecessor", storage.py:159-163)."""
    a, b = _index(a, "a"), _index(b, "b")
    if a.numel() != b.numel():
        raise ValueError("a and b must have the same length")
    keys = torch.empty_like(a)
    flag = torch.zeros(1, dtype=torch.int32, device=a.device) if check_sorted else None
    with _on(a.device):
        check(_lib.load().psa_make_keys(_ptr(a), _ptr(b), int(mul), a.numel(), _ptr(keys),
                                        _ptr(flag), _stream()))
    return keys, flag


def split_keys(keys: torch.Tensor, div: int, want_hi: bool = True, want_lo: bool = True
               ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(keys // div, keys % div): the two indices a key stream was built from,
    without a permutation gather."""
    keys = _index(keys, "keys")
    hi = torch.empty_like(keys) if want_hi else None
    lo = torch.empty_like(keys) if want_lo else None
    with _on(keys.device):
        check(_lib.load().psa_split_keys(_ptr(keys), keys.numel(), int(div), _ptr(hi), _ptr(lo),
                                         _stream()))
    return hi, lo


def needs_grad(x: Optional[torch.Tensor]) -> bool:
    """True when autograd will want a gradient for x (callers then keep the
    values on a differentiable route instead of letting them ride a sort)."""
    return x is not None and torch.is_grad_enabled() and x.requires_grad


class _GatherRows(torch.autograd.Function):
    """src[perm] with the reference's differentiability (paddle `value[perm]`,
    storage.py:166-169, transpose.py:14-22): the backward adds every output
    row's gradient back into the row it was read from — a gather through the
    inverse permutation when the caller has one, a scatter-add otherwise."""

    @staticmethod
    def forward(ctx, src, perm, inverse):
        ctx.save_for_backward(perm, inverse)
        ctx.rows = src.shape[0]
        return _gather_rows_raw(src, perm)

    @staticmethod
    def backward(ctx, grad):
        perm, inverse = ctx.saved_tensors
        grad = grad.contiguous()
        if inverse is not None:
            return _gather_rows_raw(grad, inverse), None, None
        acc = grad if grad.dtype in (torch.float32, torch.float64) else grad.float()
        return scatter(acc, perm, ctx.rows, "sum").to(grad.dtype), None, None


def gather_rows(src: torch.Tensor, perm: torch.Tensor, inverse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src[perm] along dim 0 for any dtype / trailing shape; differentiable in
    src.  `inverse` (optional): the inverse of perm when perm is a permutation
    of all rows — the backward is then a gather instead of a scatter-add."""
    if needs_grad(src):
        return _GatherRows.apply(src, perm, inverse)
    return _gather_rows_raw(src, perm)


def _gather_rows_raw(src: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    _gpu(src, "src")
    perm = _index(perm, "perm")
    src = src.contiguous()
    n = perm.numel()
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_bytes = src.element_size()
    for s in src.shape[1:]:
        row_bytes *= s
    with _on(src.device):
        check(_lib.load().psa_gather_rows(_ptr(src), _ptr(perm), n, row_bytes, _ptr(out), _stream()))
    return out


def gather_rows_window(src: torch.Tensor, perm: torch.Tensor, col0: int, width: int,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src[perm, col0:col0 + width] for a 2-D contiguous src, as a dense [n, width]
    tensor (into `out` when given): rows and feature slice packed in one pass."""
    _gpu(src, "src")
    perm = _index(perm, "perm")
    if src.dim() != 2 or not src.is_contiguous():
        raise ValueError("src must be a contiguous 2-D tensor")
    if not (0 <= col0 and col0 + width <= src.shape[1]):
        raise ValueError("column window outside src")
    n, es = perm.numel(), src.element_size()
    if out is None:
        out = torch.empty((n, width), dtype=src.dtype, device=src.device)
    elif out.shape != (n, width) or out.dtype != src.dtype or not out.is_contiguous():
        raise ValueError("out must be a contiguous [n, width] tensor of src's dtype")
    with _on(src.device):
        check(_lib.load().psa_gather_rows_window(_ptr(src), src.shape[1] * es, col0 * es, width * es, _ptr(perm), n,
                                                 _ptr(out), _stream()))
    return out


def invert_permutation(perm: torch.Tensor) -> torch.Tensor:
    perm = _index(perm, "perm")
    inv = torch.empty_like(perm)
    with _on(perm.device):
        check(_lib.load().psa_invert_permutation(_ptr(perm), perm.numel(), _ptr(inv), _stream()))
    return inv


class PermutePlan:
    """Route of every element of a 4-byte array through a fixed permutation (psa_permute_apply_u32):
    structure only, 8 bytes per element; built by permute_plan, cached by the caller."""

    def __init__(self, sl, gs, lo, n):
        self.sl, self.gs, self.lo, self.n = sl, gs, lo, n


PERMUTE_PLAN_FROM = 1 << 20  # elements from which the planned two-pass form beats n dependent 4-byte reads


def permute_plan(dest: torch.Tensor) -> PermutePlan:
    """Plan for out[i] = src[perm[i]] given dest = the INVERSE of perm (dest[s] = where source s
    goes; csr2csc / csc2csr are each other's).  Two stable sorts, one inverse, one pack pass — once
    per permutation (see include/paddle_sparse_hip.h)."""
    dest = _index(dest, "dest")
    n, dev = dest.numel(), dest.device
    if n >= (1 << 31):
        raise ValueError("permute_plan: n must be below 2^31")
    lib = _lib.load()
    T = int(lib.psa_permute_tile())
    nb = max((n + T - 1) // T, 1)
    block, _ = split_keys(dest, T, want_lo=False)
    tile, _ = split_keys(torch.arange(n, dtype=torch.int64, device=dev), T, want_lo=False)
    keys_ts, _ = make_keys(tile, block, nb)
    del tile
    _, perm_ts = index_sort(keys_ts, nb * nb, check=True)
    del keys_ts
    _, perm_mid = index_sort(block, nb, check=True)
    del block
    gslot = invert_permutation(perm_mid)
    sl = torch.empty(n, dtype=torch.int16, device=dev)
    gs = torch.empty(n, dtype=torch.int32, device=dev)
    lo = torch.empty(n, dtype=torch.int16, device=dev)
    with _on(dev):
        check(lib.psa_permute_plan_pack(_ptr(perm_ts), _ptr(gslot), _ptr(perm_mid), _ptr(dest), n, _ptr(sl), _ptr(gs),
                                        _ptr(lo), _stream()))
    return PermutePlan(sl, gs, lo, n)


def permute_apply(src: torch.Tensor, plan: PermutePlan) -> torch.Tensor:
    """src[perm] for a contiguous 1-D 4-byte array along the plan of `perm` (no autograd)."""
    _gpu(src, "src")
    if src.dim() != 1 or src.element_size() != 4 or src.numel() != plan.n or not src.is_contiguous():
        raise ValueError("permute_apply takes a contiguous 1-D array of 4-byte elements, one per planned element")
    out, mid = torch.empty_like(src), torch.empty_like(src)
    with _on(src.device):
        check(_lib.load().psa_permute_apply_u32(_ptr(src), _ptr(plan.sl), _ptr(plan.gs), _ptr(plan.lo), plan.n, _ptr(mid),
                                                _ptr(out), _stream()))
    return out


class _SegmentCsr(torch.autograd.Function):
    """segment_csr with paddle_scatter's differentiability for sum / mean (the
    coalesce and reduce(dim) call sites, storage.py:471, reduce.py:51): every
    element of a segment receives the segment's gradient (over the segment's
    length for mean); with `perm` the gradient lands on src[perm[i]]."""

    @staticmethod
    def forward(ctx, src, indptr, reduce, perm):
        if REDUCE_ID[reduce] not in (_lib.SUM, _lib.MEAN):
            raise NotImplementedError(
                f"segment_csr(reduce={reduce!r}) is not differentiable in this build (sum / mean are)")
        ctx.save_for_backward(indptr, perm)
        ctx.mean, ctx.rows = REDUCE_ID[reduce] == _lib.MEAN, src.shape[0]
        return _segment_csr_raw(src, indptr, reduce, perm)

    @staticmethod
    def backward(ctx, grad):
        indptr, perm = ctx.saved_tensors
        n = perm.numel() if perm is not None else ctx.rows
        grad = grad.contiguous()
        if ctx.mean:
            count = (indptr[1:] - indptr[:-1]).clamp(min=1).to(grad.dtype)
            grad = grad / count.view((-1,) + (1,) * (grad.dim() - 1))
        g = _gather_rows_raw(grad, ptr2ind(indptr, n))  # the segment's gradient for each of its elements
        if perm is not None:
            acc = g if g.dtype in (torch.float32, torch.float64) else g.float()
            g = scatter(acc, perm, ctx.rows, "sum").to(grad.dtype)
        return g, None, None, None


def segment_csr(src: torch.Tensor, indptr: torch.Tensor, reduce: str = "sum",
                perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """paddle_scatter.segment_csr(src, indptr, reduce=...) along dim 0
    (call sites storage.py:471, reduce.py:51); with `perm`, reduces
    src[perm] without materialising it.  Differentiable in src for sum / mean."""
    if needs_grad(src):
        return _SegmentCsr.apply(src, indptr, reduce, perm)
    return _segment_csr_raw(src, indptr, reduce, perm)


def _segment_csr_raw(src: torch.Tensor, indptr: torch.Tensor, reduce: str = "sum",
                     perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    _gpu(src, "src")
    indptr = _index(indptr, "indptr")
    if src.dtype not in _DTYPE_ID:
        raise TypeError(f"segment_csr: unsupported dtype {src.dtype}")
    src = src.contiguous()
    if perm is not None:
        perm = _index(perm, "perm")
    nseg = indptr.numel() - 1
    D = 1
    for s in src.shape[1:]:
        D *= s
    n_rows = perm.numel() if perm is not None else src.shape[0]
    out = torch.empty((nseg,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    with _on(src.device):
        check(_lib.load().psa_segment_reduce(REDUCE_ID[reduce], _DTYPE_ID[src.dtype], _ptr(src),
                                             _ptr(perm), _ptr(indptr), nseg, D, n_rows,
                                             _ptr(out), _stream()))
    return out


def _unique_count(sorted_keys: torch.Tensor, after: Optional["SortScratch"]):
    """Phase 1 of unique_sorted: run heads counted per block and scanned; the ONE host read of the count (with the
    preceding sort's look-back diagnostic when `after` is given).  Returns (count, scratch, device count word)."""
    n, dev = sorted_keys.numel(), sorted_keys.device
    lib = _lib.load()
    ws = _workspace(lib.psa_unique_workspace_bytes(n), dev)
    count_d = torch.empty(2, dtype=torch.int64, device=dev)
    with _on(dev):
        if after is not None and after.n == n:
            check(lib.psa_unique_count_after_sort(_ptr(sorted_keys), n, _ptr(ws), ws.numel(), _ptr(after.ws),
                                                  after.max_value, _ptr(count_d), _stream()))
            count, fault = count_d.tolist()
            if fault:
                raise HipCoreError("index_sort: an inter-workgroup wait of a radix pass gave up; the order is invalid")
        else:
            check(lib.psa_unique_count(_ptr(sorted_keys), n, _ptr(ws), ws.numel(), _ptr(count_d), _stream()))
            count = int(count_d[0].item())
    return count, ws, count_d


def unique_sorted(sorted_keys: torch.Tensor, N: int, want_ptr: bool = True,
                  want_rowcol: bool = True, after: Optional[SortScratch] = None, _counted=None):
    """Run-length structure of SORTED keys (storage.py:455-470 without the
    bool-mask selects).  Returns (count, ptr | None, row | None, col | None)
    with row = key // N, col = key % N of each distinct key.  One host sync
    (reading the count to size the outputs) — the reference has three.
    after: the SortScratch of the sort that produced the keys; its look-back
    diagnostic comes back with the count (HipCoreError if a wait gave up)."""
    sorted_keys = _index(sorted_keys, "sorted_keys")
    n = sorted_keys.numel()
    dev = sorted_keys.device
    lib = _lib.load()
    count, ws, count_d = _counted if _counted is not None else _unique_count(sorted_keys, after)
    with _on(dev):
        ptr = torch.empty(count + 1, dtype=torch.int64, device=dev) if want_ptr else None
        # row and col are the two rows of ONE [2, count] buffer, so the
        # functional API's `stack([row, col])` (coalesce.py:29) is `row._base`
        index = torch.empty((2, count), dtype=torch.int64, device=dev) if want_rowcol else None
        row = index[0] if want_rowcol else None
        col = index[1] if want_rowcol else None
        if n > 0 and (want_ptr or want_rowcol):
            check(lib.psa_unique_write(_ptr(sorted_keys), n, int(N), _ptr(ws), _ptr(count_d),
                                       _ptr(ptr), _ptr(row), _ptr(col), _stream()))
        elif want_ptr:
            ptr.zero_()
    return count, ptr, row, col


def unique_sorted_reduce(sorted_keys: torch.Tensor, N: int, payload: torch.Tensor, reduce: str = "sum",
                         after: Optional[SortScratch] = None):
    """unique_sorted + segment_csr for 4-byte scalar values that are in the keys' sorted order already (they rode the
    sort as its payload, or the input was sorted): returns (count, row, col, value') with value'[s] = the reduction of
    run s.  When the runs are short (count * 32 > n) the index and the reduced values come from ONE launch and no
    ptr array is written (psa_unique_write_reduce); otherwise the two-launch form, whose reducer takes a wave per run."""
    sorted_keys = _index(sorted_keys, "sorted_keys")
    _gpu(payload, "payload")
    n, dev = sorted_keys.numel(), sorted_keys.device
    if payload.dim() != 1 or payload.numel() != n or payload.dtype not in (torch.float32, torch.int32):
        raise ValueError("payload must be float32[n] or int32[n]")
    counted = _unique_count(sorted_keys, after)
    count, ws, count_d = counted
    if 0 < count < n and count * 32 > n:
        payload = payload.contiguous()
        index = torch.empty((2, count), dtype=torch.int64, device=dev)
        value = torch.empty(count, dtype=payload.dtype, device=dev)
        with _on(dev):
            check(_lib.load().psa_unique_write_reduce(REDUCE_ID[reduce], _DTYPE_ID[payload.dtype], _ptr(sorted_keys), n, int(N),
                                                      _ptr(ws), _ptr(count_d), _ptr(index), _ptr(payload), _ptr(value), _stream()))
        return count, index[0], index[1], value
    count, ptr, row, col = unique_sorted(sorted_keys, N, want_ptr=count < n, _counted=counted)
    return count, row, col, (payload if count == n else _segment_csr_raw(payload, ptr, reduce))


# ---------------------------------------------------------------------------
# SpMM backward
# ---------------------------------------------------------------------------

def _f32(x: torch.Tensor, name: str) -> torch.Tensor:
    _gpu(x, name)
    if x.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {x.dtype})")
    return x.contiguous()


def spmm_value_bw(row, rowptr, col, mat, grad, reduce: str = "sum") -> torch.Tensor:
    """Upstream spmm_value_bw(row, rowptr, col, mat, grad, reduce): gradient
    wrt the nnz values for sum/mean.  `row` is accepted for signature parity
    but not read (one wave per CSR row knows its row)."""
    rowptr, col = _index(rowptr, "rowptr"), _index(col, "col")
    mat, grad = _f32(mat, "mat"), _f32(grad, "grad")
    if reduce not in ("sum", "add", "mean"):
        raise ValueError("spmm_value_bw: reduce must be sum or mean")
    M, K, nnz = rowptr.numel() - 1, mat.shape[1], col.numel()
    if grad.shape != (M, K):
        raise ValueError("grad must be [M, K]")
    out = torch.empty(nnz, dtype=torch.float32, device=mat.device)
    lib = _lib.load()
    ws_bytes = lib.psa_spmm_value_bw_workspace_bytes(nnz)
    ws = _workspace(ws_bytes, mat.device) if ws_bytes else None
    with _on(mat.device):
        check(lib.psa_spmm_value_bw(REDUCE_ID[reduce], _ptr(rowptr), _ptr(col), _ptr(mat),
                                    _ptr(grad), M, K, nnz, _ptr(out), _ptr(ws), ws_bytes, _stream()))
    return out


def transpose_weights(value, csr2csc, row_csc, rowptr, mean: bool) -> torch.Tensor:
    """Edge weights in CSC order for gB = A^T gOut (value[csr2csc], scaled by
    1/deg(row) for mean)."""
    csr2csc = _index(csr2csc, "csr2csc")
    nnz = csr2csc.numel()
    if value is not None:
        value = _f32(value, "value")
    if mean:
        row_csc, rowptr = _index(row_csc, "row_csc"), _index(rowptr, "rowptr")
    out = torch.empty(nnz, dtype=torch.float32, device=csr2csc.device)
    with _on(csr2csc.device):
        check(_lib.load().psa_transpose_weights(_ptr(value), _ptr(csr2csc),
                                                _ptr(row_csc) if mean else None,
                                                _ptr(rowptr) if mean else None,
                                                nnz, int(mean), _ptr(out), _stream()))
    return out


def spmm_minmax_bw(col, value, mat, grad, arg_out, want_value: bool = True, want_mat: bool = True):
    """Backward of spmm_min / spmm_max through arg_out.  Returns
    (grad_value | None, grad_mat | None)."""
    col = _index(col, "col")
    mat, grad = _f32(mat, "mat"), _f32(grad, "grad")
    _gpu(arg_out, "arg_out")
    arg_out = arg_out.contiguous()
    if value is not None:
        value = _f32(value, "value")
    (N, K), M, nnz = mat.shape, grad.shape[0], col.numel()
    gv = torch.empty(nnz, dtype=torch.float32, device=mat.device) if want_value else None
    gm = torch.empty((N, K), dtype=torch.float32, device=mat.device) if want_mat else None
    with _on(mat.device):
        check(_lib.load().psa_spmm_minmax_bw(_ptr(col), _ptr(value), _ptr(mat), _ptr(grad),
                                             _ptr(arg_out), M, N, K, nnz, _ptr(gv), _ptr(gm),
                                             _stream()))
    return gv, gm


def csc_edge_tags(rowptr, row_csc, csr2csc, width: int = 1) -> torch.Tensor:
    """Position of every CSC-ordered edge inside its CSR row, in the form of arg_bytes:
    width 1 -> uint8[nnz] (mod 128; bit 7 marks rows of more than 128 edges), width 2 ->
    int16[nnz] (mod 65 536).  Structure only — cache it next to csr2csc."""
    if width not in (1, 2):
        raise ValueError("width must be 1 or 2")
    rowptr, row_csc, csr2csc = _index(rowptr, "rowptr"), _index(row_csc, "row_csc"), _index(csr2csc, "csr2csc")
    tag = torch.empty(csr2csc.numel(), dtype=torch.int16 if width == 2 else torch.uint8, device=csr2csc.device)
    with _on(csr2csc.device):
        check(_lib.load().psa_csc_edge_tags(_ptr(rowptr), _ptr(row_csc), _ptr(csr2csc),
                                            csr2csc.numel(), _ptr(tag), width, _stream()))
    return tag


def minmax_bw_csc_supported(K: int) -> bool:
    return K % 4 == 0 and 0 < K <= 256


def spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, tag, value, mat, grad, arg_out,
                       want_value: bool = True, csc2csr: Optional[torch.Tensor] = None,
                       arg_bytes: Optional[torch.Tensor] = None, hot_ids: Optional[torch.Tensor] = None,
                       to_csr_plan: Optional["PermutePlan"] = None, value_csc: Optional[torch.Tensor] = None,
                       hot_bytes_tail: Optional[torch.Tensor] = None):
    """Backward of spmm_min / spmm_max in one pass over the CSC view, no atomics
    (see include/paddle_sparse_hip.h).  Returns (grad_value f32[nnz] | None,
    grad_mat f32[N, K]); grad_value is in CSR order (the pass writes it in CSC
    order, `csc2csr` — computed here when not given — brings it back).
    tag / arg_bytes: uint8 (one byte per entry) or int16 (two).  hot_ids: int64[h]
    rows of grad / arg_bytes that row_csc refers to as M + position (compact copies
    of them are gathered here; needs arg_bytes exact, arg_out None).  hot_bytes_tail: int16[t, K], the
    row-local winners of the LAST t rows hot_ids names, given instead of gathered (the pieces of rows above
    65 535 entries, SparseStorage._huge_backward_plan)."""
    rowptr, colptr = _index(rowptr, "rowptr"), _index(colptr, "colptr")
    row_csc, csr2csc = _index(row_csc, "row_csc"), _index(csr2csc, "csr2csc")
    grad = _f32(grad, "grad")
    if arg_out is None and arg_bytes is None:
        raise ValueError("spmm_minmax_bw_csc needs arg_out or arg_bytes")
    if arg_out is not None:
        _gpu(arg_out, "arg_out")
        arg_out = arg_out.contiguous()
    if value is not None:
        value = _f32(value, "value")
    edge_ids = csr2csc
    if value_csc is not None:
        if arg_out is not None:
            raise ValueError("value_csc goes with the exact arg_bytes forms (arg_out is compared with CSR edge ids)")
        value, edge_ids = _f32(value_csc, "value_csc"), None
    _gpu(tag, "tag")
    if tag.dtype not in (torch.uint8, torch.int16) or tag.numel() != csr2csc.numel():
        raise ValueError("tag must be uint8[nnz] or int16[nnz] (ops.csc_edge_tags)")
    width = tag.element_size()
    (M, K), N, nnz = grad.shape, colptr.numel() - 1, csr2csc.numel()
    if arg_out is not None and (arg_out.shape != grad.shape or arg_out.dtype != torch.int64):
        raise ValueError("arg_out must be int64[M, K] like grad")
    if arg_bytes is not None:
        _gpu(arg_bytes, "arg_bytes")
        if arg_bytes.dtype != tag.dtype or arg_bytes.shape != grad.shape or not arg_bytes.is_contiguous():
            raise ValueError("arg_bytes must be a contiguous [M, K] tensor of the tags' dtype (the third result of ops._spmm)")
    hot_grad = hot_bytes = None
    num_hot = 0
    if hot_ids is not None and hot_ids.numel():
        if arg_bytes is None or arg_out is not None or width != 2:
            raise ValueError("hot_ids need an exact arg_bytes in the two-byte form and no arg_out")
        hot_ids = _index(hot_ids, "hot_ids")
        num_hot = hot_ids.numel()
        hot_grad, hot_bytes = _gather_rows_raw(grad, hot_ids), _hot_bytes(arg_bytes, hot_ids, hot_bytes_tail)
    gv = None
    if want_value:
        mat = _f32(mat, "mat")
        if mat.shape != (N, K):
            raise ValueError("mat must be [N, K]")
        gv = torch.empty(nnz, dtype=torch.float32, device=grad.device)
    gm = torch.empty((N, K), dtype=torch.float32, device=grad.device)
    lib = _lib.load()
    ws = _workspace(lib.psa_spmm_minmax_bw_csc_workspace_bytes(M, K, nnz), grad.device)
    with _on(grad.device):
        check(lib.psa_spmm_minmax_bw_csc(_ptr(rowptr), _ptr(colptr), _ptr(row_csc), _ptr(edge_ids),
                                         _ptr(tag.contiguous()), _ptr(value),
                                         _ptr(mat) if want_value else None, _ptr(grad), _ptr(arg_out),
                                         _ptr(arg_bytes), width, _ptr(hot_grad), _ptr(hot_bytes), num_hot,
                                         M, N, K, nnz, _ptr(gv), _ptr(gm), _ptr(ws), ws.numel(), _stream()))
    if gv is not None:
        if to_csr_plan is not None:
            gv = permute_apply(gv, to_csr_plan)
        else:
            gv = gather_rows(gv, csc2csr if csc2csr is not None else invert_permutation(csr2csc))
    return gv, gm


def _hot_bytes(arg_bytes: torch.Tensor, hot_ids: torch.Tensor, tail: Optional[torch.Tensor]) -> torch.Tensor:
    """Compact copy of the rows `hot_ids` of arg_bytes; the last len(tail) rows are `tail` itself."""
    if tail is None or tail.shape[0] == 0:
        return _gather_rows_raw(arg_bytes, hot_ids)
    _gpu(tail, "hot_bytes_tail")
    t = tail.shape[0]
    if tail.dtype != arg_bytes.dtype or tail.dim() != 2 or tail.shape[1] != arg_bytes.shape[1] or t > hot_ids.numel():
        raise ValueError("hot_bytes_tail must be [t <= len(hot_ids), K] of arg_bytes' dtype")
    out = torch.empty((hot_ids.numel(), arg_bytes.shape[1]), dtype=arg_bytes.dtype, device=arg_bytes.device)
    head = hot_ids.numel() - t
    if head:
        out[:head] = _gather_rows_raw(arg_bytes, hot_ids[:head])
    out[head:] = tail
    return out


def spmm_minmax_bw_eb(colptr, col_csc, row_csc, tag, weight_csc, grad, arg_bytes, hot_ids=None,
                      hot_bytes_tail: Optional[torch.Tensor] = None) -> torch.Tensor:
    """grad_mat f32[N, K] of spmm_min / spmm_max for a fixed adjacency, by the edge-range kernels over
    the CSC view (psa_spmm_minmax_bw_eb): power-law matrices.  col_csc: column of every CSC-ordered
    entry or None; tag / arg_bytes: uint8 or int16 (exact form); weight_csc: value[csr2csc] or None;
    hot_ids: rows of grad / arg_bytes that row_csc names as M + position (copies gathered here)."""
    colptr, row_csc = _index(colptr, "colptr"), _index(row_csc, "row_csc")
    if col_csc is not None:
        col_csc = _index(col_csc, "col_csc")
    grad = _f32(grad, "grad")
    _gpu(tag, "tag")
    _gpu(arg_bytes, "arg_bytes")
    if tag.dtype not in (torch.uint8, torch.int16) or arg_bytes.dtype != tag.dtype or arg_bytes.shape != grad.shape:
        raise ValueError("tag / arg_bytes must be uint8 or int16, arg_bytes [M, K] like grad")
    arg_bytes, tag = arg_bytes.contiguous(), tag.contiguous()
    if weight_csc is not None:
        weight_csc = _f32(weight_csc, "weight_csc")
    (M, K), N, nnz = grad.shape, colptr.numel() - 1, row_csc.numel()
    hot_grad = hot_bytes = None
    num_hot = 0
    if hot_ids is not None and hot_ids.numel():
        hot_ids = _index(hot_ids, "hot_ids")
        num_hot = hot_ids.numel()
        hot_grad, hot_bytes = _gather_rows_raw(grad, hot_ids), _hot_bytes(arg_bytes, hot_ids, hot_bytes_tail)
    gm = torch.empty((N, K), dtype=torch.float32, device=grad.device)
    lib = _lib.load()
    ws = _workspace(lib.psa_spmm_minmax_bw_eb_workspace_bytes(K, nnz), grad.device)
    with _on(grad.device):
        check(lib.psa_spmm_minmax_bw_eb(_ptr(colptr), _ptr(col_csc), _ptr(row_csc), _ptr(tag), _ptr(weight_csc), _ptr(grad),
                                        _ptr(arg_bytes), tag.element_size(), _ptr(hot_grad), _ptr(hot_bytes), num_hot,
                                        M, N, K, nnz, _ptr(gm), _ptr(ws), ws.numel(), _stream()))
    return gm


def spmm_sum_bw_csc(colptr, row_csc, csr2csc, value, mat, grad, want_value: bool = True,
                    csc2csr: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
                    hot_ids: Optional[torch.Tensor] = None, to_csr_plan: Optional["PermutePlan"] = None,
                    value_csc: Optional[torch.Tensor] = None):
    """sum backward, both gradients in one pass over the CSC view (see
    include/paddle_sparse_hip.h).  Returns (grad_value f32[nnz] | None, in CSR
    order, grad_mat f32[N, K]).  For mean, pass row_scale = 1 / max(deg, 1)
    (f32[M]): it multiplies both gradients per edge.  hot_ids: int64[h] rows of grad
    that row_csc refers to as M + position (a compact copy is gathered here)."""
    colptr, row_csc, csr2csc = _index(colptr, "colptr"), _index(row_csc, "row_csc"), _index(csr2csc, "csr2csc")
    grad = _f32(grad, "grad")
    if value is not None:
        value = _f32(value, "value")
    (M, K), N, nnz = grad.shape, colptr.numel() - 1, csr2csc.numel()
    if value_csc is not None:  # value[csr2csc] at hand (SparseStorage._permute_plan("to_csc")): the pass reads it as a stream
        value, edge_ids = _f32(value_csc, "value_csc"), None
    else:
        edge_ids = csr2csc
    hot_grad, num_hot = None, 0
    if hot_ids is not None and hot_ids.numel():
        hot_ids = _index(hot_ids, "hot_ids")
        num_hot = hot_ids.numel()
        hot_grad = _gather_rows_raw(grad, hot_ids)
    if row_scale is not None:
        row_scale = _f32(row_scale, "row_scale")
        if row_scale.shape != (M,):
            raise ValueError("row_scale must be f32[M]")
        if num_hot:
            row_scale = torch.cat([row_scale, row_scale[hot_ids]])
    gv = None
    if want_value:
        mat = _f32(mat, "mat")
        if mat.shape != (N, K):
            raise ValueError("mat must be [N, K]")
        gv = torch.empty(nnz, dtype=torch.float32, device=grad.device)
    gm = torch.empty((N, K), dtype=torch.float32, device=grad.device)
    lib = _lib.load()
    ws = _workspace(lib.psa_spmm_sum_bw_csc_workspace_bytes(K, nnz), grad.device)
    with _on(grad.device):
        check(lib.psa_spmm_sum_bw_csc(_ptr(colptr), _ptr(row_csc), _ptr(edge_ids), _ptr(value),
                                      _ptr(row_scale), _ptr(mat) if want_value else None, _ptr(grad),
                                      _ptr(hot_grad), num_hot, M, N, K, nnz,
                                      _ptr(gv), _ptr(gm), _ptr(ws), ws.numel(), _stream()))
    if gv is not None:
        if to_csr_plan is not None:
            gv = permute_apply(gv, to_csr_plan)
        else:
            gv = gather_rows(gv, csc2csr if csc2csr is not None else invert_permutation(csr2csc))
    return gv, gm


def half_sum_bw_csc_supported(K: int) -> bool:
    return K % 8 == 0 and 0 < K <= 512


def _half_long_workspace(K: int, nnz: int, device, long_columns: bool):
    """Scratch of the long-column path of the half-width passes over the CSC view (chunk list + fp32 partials)."""
    if not long_columns:
        return None
    nbytes = _lib.load().psa_spmm_half_bw_csc_workspace_bytes(K, nnz)
    return _workspace(nbytes, device) if nbytes else None


def spmm_half_sum_bw_csc(colptr, row_csc, weight_csc, mat, grad, want_value: bool = True,
                         row_scale: Optional[torch.Tensor] = None, long_columns: bool = True):
    """sum / mean backward over the CSC view with fp16 / bf16 dense operands (psa_spmm_half_sum_bw_csc):
    returns (grad_value f32[nnz] IN CSC ORDER | None, grad_mat [N, K] in grad's dtype).  weight_csc:
    f32[nnz] = value[csr2csc] or None; row_scale f32[M] (mean) or None.  The caller brings
    grad_value to CSR order (SparseStorage._permute_plan("to_csr") / csc2csr).  long_columns=False: the caller
    knows that no column has more than 128 entries (the view's _longest_row()) and saves the two near-empty launches."""
    colptr, row_csc = _index(colptr, "colptr"), _index(row_csc, "row_csc")
    _gpu(grad, "grad")
    if grad.dtype not in (torch.float16, torch.bfloat16) or grad.dim() != 2:
        raise TypeError("grad must be a 2-D float16 / bfloat16 tensor")
    grad = grad.contiguous()
    (M, K), N, nnz = grad.shape, colptr.numel() - 1, row_csc.numel()
    if weight_csc is not None:
        weight_csc = _f32(weight_csc, "weight_csc")
    if row_scale is not None:
        row_scale = _f32(row_scale, "row_scale")
        if row_scale.shape != (M,):
            raise ValueError("row_scale must be f32[M]")
    gv = None
    if want_value:
        _gpu(mat, "mat")
        if mat.dtype != grad.dtype or mat.shape != (N, K):
            raise ValueError("mat must be [N, K] in grad's dtype")
        mat = mat.contiguous()
        gv = torch.empty(nnz, dtype=torch.float32, device=grad.device)
    gm = torch.empty((N, K), dtype=grad.dtype, device=grad.device)
    ws = _half_long_workspace(K, nnz, grad.device, long_columns)
    with _on(grad.device):
        check(_lib.load().psa_spmm_half_sum_bw_csc(_DTYPE_ID[grad.dtype], _ptr(colptr), _ptr(row_csc), _ptr(weight_csc),
                                                   _ptr(row_scale), _ptr(mat) if want_value else None, _ptr(grad), M, N, K,
                                                   nnz, _ptr(gv), _ptr(gm), _ptr(ws), ws.numel() if ws is not None else 0,
                                                   _stream()))
    return gv, gm


def spmm_half_minmax_bw_csc(colptr, row_csc, tag, weight_csc, mat, grad, arg_bytes, want_value: bool = True,
                            long_columns: bool = True):
    """min / max backward over the CSC view with fp16 / bf16 dense operands (psa_spmm_half_minmax_bw_csc):
    returns (grad_value f32[nnz] IN CSC ORDER | None, grad_mat [N, K] in grad's dtype).  tag / arg_bytes:
    uint8 or int16, the exact row-local forms (ops.csc_edge_tags / the third result of ops._spmm)."""
    colptr, row_csc = _index(colptr, "colptr"), _index(row_csc, "row_csc")
    _gpu(grad, "grad")
    if grad.dtype not in (torch.float16, torch.bfloat16) or grad.dim() != 2:
        raise TypeError("grad must be a 2-D float16 / bfloat16 tensor")
    grad = grad.contiguous()
    (M, K), N, nnz = grad.shape, colptr.numel() - 1, row_csc.numel()
    _gpu(tag, "tag")
    _gpu(arg_bytes, "arg_bytes")
    if tag.dtype not in (torch.uint8, torch.int16) or arg_bytes.dtype != tag.dtype or arg_bytes.shape != grad.shape or tag.numel() != nnz:
        raise ValueError("tag / arg_bytes must be uint8 or int16, tag [nnz], arg_bytes [M, K] like grad")
    arg_bytes, tag = arg_bytes.contiguous(), tag.contiguous()
    if weight_csc is not None:
        weight_csc = _f32(weight_csc, "weight_csc")
    gv = None
    if want_value:
        _gpu(mat, "mat")
        if mat.dtype != grad.dtype or mat.shape != (N, K):
            raise ValueError("mat must be [N, K] in grad's dtype")
        mat = mat.contiguous()
        gv = torch.empty(nnz, dtype=torch.float32, device=grad.device)
    gm = torch.empty((N, K), dtype=grad.dtype, device=grad.device)
    ws = _half_long_workspace(K, nnz, grad.device, long_columns)
    with _on(grad.device):
        check(_lib.load().psa_spmm_half_minmax_bw_csc(_DTYPE_ID[grad.dtype], _ptr(colptr), _ptr(row_csc), _ptr(tag),
                                                      _ptr(weight_csc), _ptr(mat) if want_value else None, _ptr(grad),
                                                      _ptr(arg_bytes), tag.element_size(), M, N, K, nnz, _ptr(gv), _ptr(gm),
                                                      _ptr(ws), ws.numel() if ws is not None else 0, _stream()))
    return gv, gm


def bincount(index: torch.Tensor, size: int) -> torch.Tensor:
    """int64[size] occurrence counts (colcount, storage.py:414-418)."""
    index = _index(index, "index")
    out = torch.empty(size, dtype=torch.int64, device=index.device)
    with _on(index.device):
        check(_lib.load().psa_bincount(_ptr(index), index.numel(), size, _ptr(out), _stream()))
    return out


def count2ptr(counts: torch.Tensor) -> torch.Tensor:
    """[0, cumsum(counts)] as int64[n+1] (colptr, storage.py:397-398)."""
    counts = _index(counts, "counts")
    n = counts.numel()
    out = torch.empty(n + 1, dtype=torch.int64, device=counts.device)
    lib = _lib.load()
    ws = _workspace(lib.psa_count2ptr_workspace_bytes(n), counts.device)
    with _on(counts.device):
        check(lib.psa_count2ptr(_ptr(counts), n, _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


def scatter(src: torch.Tensor, index: torch.Tensor, dim_size: int, reduce: str = "sum") -> torch.Tensor:
    """paddle_scatter.scatter(src, index, 0, None, dim_size, reduce) as called
    by reduce.py:42 (reduction over sparse dim 0, index = col)."""
    _gpu(src, "src")
    index = _index(index, "index")
    if src.dtype not in (torch.float32, torch.float64, torch.int32, torch.int64):
        raise TypeError(f"scatter: unsupported dtype {src.dtype}")
    src = src.contiguous()
    if src.shape[0] != index.numel():
        raise ValueError("src.shape[0] must equal index.numel()")
    D = 1
    for s in src.shape[1:]:
        D *= s
    out = torch.empty((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    lib = _lib.load()
    ws = _workspace(lib.psa_scatter_workspace_bytes(dim_size), src.device)
    with _on(src.device):
        check(lib.psa_scatter_reduce(REDUCE_ID[reduce], _DTYPE_ID[src.dtype], _ptr(src), _ptr(index),
                                     index.numel(), D, dim_size, _ptr(out), _ptr(ws), ws.numel(),
                                     _stream()))
    return out


def spspmm_count(colA: torch.Tensor, rowptrB: torch.Tensor) -> torch.Tensor:
    """counts[e] = stored entries of B's row colA[e] (products A entry e takes part in)."""
    colA, rowptrB = _index(colA, "colA"), _index(rowptrB, "rowptrB")
    out = torch.empty_like(colA)
    with _on(colA.device):
        check(_lib.load().psa_spspmm_count(_ptr(colA), colA.numel(), _ptr(rowptrB), _ptr(out),
                                           _stream()))
    return out


def spspmm_expand(rowA, colA, valA, rowptrB, colB, valB, offsets, owner, total: int, n: int,
                  dtype: torch.dtype):
    """All `total` partial products of A @ B as (key = i * n + j, value) pairs,
    in A-storage order (see csrc/spspmm.hip)."""
    rowA, colA = _index(rowA, "rowA"), _index(colA, "colA")
    rowptrB, colB = _index(rowptrB, "rowptrB"), _index(colB, "colB")
    offsets, owner = _index(offsets, "offsets"), _index(owner, "owner")
    if dtype not in (torch.float32, torch.float64, torch.int32, torch.int64):
        raise TypeError(f"spspmm: unsupported dtype {dtype}")
    for name, v, cnt in (("valueA", valA, colA.numel()), ("valueB", valB, colB.numel())):
        if v is not None:
            _gpu(v, name)
            if v.dtype != dtype or v.dim() != 1 or v.numel() != cnt or not v.is_contiguous():
                raise ValueError(f"{name} must be a contiguous 1-D {dtype} tensor with one entry per index")
    has_value = valA is not None or valB is not None
    keys = torch.empty(total, dtype=torch.int64, device=colA.device)
    vals = torch.empty(total, dtype=dtype, device=colA.device) if has_value else None
    with _on(colA.device):
        check(_lib.load().psa_spspmm_expand(_DTYPE_ID[dtype], _ptr(rowA), _ptr(colA), _ptr(valA),
                                            _ptr(rowptrB), _ptr(colB), _ptr(valB), _ptr(offsets),
                                            _ptr(owner), int(total), int(n), _ptr(keys), _ptr(vals),
                                            _stream()))
    return keys, vals


def sample_adj(rowptr: torch.Tensor, col: torch.Tensor, idx: torch.Tensor, num_neighbors: int,
               replace: bool = False, seed: int = 0, num_cols: Optional[int] = None
               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """paddle_sparse_ops.sample_adj(rowptr, col, idx, num_neighbors, replace)
    (csrc/sample.cpp:8-24, CPU text csrc/cpu/sample_cpu.cpp:9-148) on GPU
    tensors: returns (out_rowptr, out_col, n_id, e_id).  `seed` selects the
    counter-based random stream (see include/paddle_sparse_hip.h); `num_cols`
    (exclusive bound on col, known to the SparseTensor caller) saves the
    col.max() pass that sizes the relabel scratch."""
    rowptr, col, idx = _index(rowptr, "rowptr"), _index(col, "col"), _index(idx, "idx")
    dev = rowptr.device
    lib = _lib.load()
    S, k, rep, seed = idx.numel(), int(num_neighbors), int(bool(replace)), int(seed) & (2**64 - 1)
    num_nodes = rowptr.numel() - 1
    with _on(dev):
        counts = torch.empty(S, dtype=torch.int64, device=dev)
        check(lib.psa_sample_count(_ptr(rowptr), _ptr(idx), S, k, rep, _ptr(counts), _stream()))
        out_rowptr = count2ptr(counts)
        E = int(out_rowptr[-1].item())
        owner = ptr2ind(out_rowptr, E)
        e_raw = torch.empty(E, dtype=torch.int64, device=dev)
        check(lib.psa_sample_select(_ptr(rowptr), _ptr(idx), S, _ptr(out_rowptr), _ptr(owner), E,
                                    k, rep, seed, _ptr(e_raw), _stream()))
        # node ids live in [0, max(num_rows, num_cols)): col values index newid too
        if num_cols is not None:
            n_scratch = max(num_nodes, int(num_cols))
        else:
            n_scratch = max(num_nodes, int(col.max().item()) + 1 if col.numel() else 0) if E else num_nodes
        newid = torch.empty(n_scratch, dtype=torch.int64, device=dev)
        flags = torch.empty(E, dtype=torch.int64, device=dev)
        check(lib.psa_relabel_mark(_ptr(idx), S, _ptr(col), _ptr(e_raw), E, n_scratch,
                                   _ptr(newid), _ptr(flags), _stream()))
        rank = count2ptr(flags)
        n_out = S + int(rank[-1].item())
        n_id = torch.empty(n_out, dtype=torch.int64, device=dev)
        keys = torch.empty(E, dtype=torch.int64, device=dev)
        check(lib.psa_relabel_finish(_ptr(idx), S, _ptr(col), _ptr(e_raw), E, _ptr(newid),
                                     _ptr(rank), _ptr(owner), n_out, _ptr(n_id), _ptr(keys),
                                     _stream()))
    if E == 0:
        return out_rowptr, keys, n_id, e_raw
    keys, perm = index_sort(keys, max(S * n_out, 1), with_sorted_inputs=True, check=True)
    _, out_col = split_keys(keys, n_out, want_hi=False)
    return out_rowptr, out_col, n_id, gather_rows(e_raw, perm)
