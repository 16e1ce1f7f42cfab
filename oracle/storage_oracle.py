"""TEST INFRASTRUCTURE — CPU oracle, never part of the shipped path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

numpy restatement of the reference's Python hot path (the reference runs it
on Paddle ops + paddle_scatter, neither importable here):

  ctor sort ............ /root/reference/paddle_sparse/storage.py:158-171
  index_sort ........... /root/reference/paddle_sparse/utils.py:14-23
  rowcount/colptr/colcount ... storage.py:373-420
  csr2csc / csc2csr .... storage.py:425-447
  is_coalesced/coalesce  storage.py:449-486
  coalesce() ........... /root/reference/paddle_sparse/coalesce.py:25-29
  t() / transpose() .... /root/reference/paddle_sparse/transpose.py:9-33,41-65
  reduction() .......... /root/reference/paddle_sparse/reduce.py:12-71
  to_symmetric() ....... /root/reference/paddle_sparse/tensor.py:415-451
  add() / mul() ........ /root/reference/paddle_sparse/add.py:30-47, mul.py:35-73
  index_select / masked_select / narrow (and the _nnz forms)
                         /root/reference/paddle_sparse/index_select.py:13-101,
                         masked_select.py:12-96, narrow.py:11-100

Third-party arithmetic that is not under /root/reference:
  paddle_scatter.segment_csr / scatter / scatter_add (unpinned HEAD,
  README.md:21-24).  Restated with pytorch_scatter's published semantics:
  empty segment -> 0, mean = sum / max(count, 1) (floor division for integer
  dtypes), min/max return values only.
  paddle.argsort (utils.py:22, stable=False by default): for duplicate keys
  the permutation is not unique; this oracle (and the HIP radix sort) use the
  STABLE permutation, which is one of the valid answers and the only
  reproducible one.

Pinned by the reference's known answers (tests/golden/reference_kats.json,
from test/test_storage.py, test/test_coalesce.py, test/test_transpose.py,
test/test_tensor.py, test/test_eye.py, test/test_add.py, test/test_mul.py,
README.md:204-264).  reduce(dim=0/1), mean/min coalesce: no reference fixture
(test/test_reduce.py only covers dim=None) — pinned instead by the third-party
answers frozen in tests/golden/third_party.npz (numpy ufunc.at,
torch.segment_reduce; tests/golden/make_golden.py).  The slicing functions:
the reference's own tests are shape-only (test/test_tensor.py:16-68, in
reference_kats.json as "getitem_shapes"); values are checked against dense
numpy indexing of the same matrix.
"""
from __future__ import annotations

import numpy as np

REDUCTIONS = ("sum", "add", "mean", "min", "max")


def index_sort(keys: np.ndarray) -> np.ndarray:
    """utils.py:22 `inputs.argsort()` under the stable convention."""
    return np.argsort(np.asarray(keys, dtype=np.int64), kind="stable").astype(np.int64)


def _is_int(a: np.ndarray) -> bool:
    return np.issubdtype(a.dtype, np.integer)


def segment_csr(src: np.ndarray, indptr: np.ndarray, reduce: str = "sum") -> np.ndarray:
    """pytorch_scatter.segment_csr along dim 0 (call sites storage.py:471,
    reduce.py:51).  Accumulates in segment order in src's dtype."""
    assert reduce in REDUCTIONS
    src = np.asarray(src)
    indptr = np.asarray(indptr, dtype=np.int64)
    nseg = indptr.size - 1
    out = np.zeros((nseg,) + src.shape[1:], dtype=src.dtype)
    for i in range(nseg):
        lo, hi = int(indptr[i]), int(indptr[i + 1])
        if hi <= lo:
            continue
        seg = src[lo:hi]
        if reduce in ("sum", "add", "mean"):
            acc = seg[0].copy()
            for j in range(1, hi - lo):
                acc = acc + seg[j]
            if reduce == "mean":
                acc = acc // (hi - lo) if _is_int(src) else (acc / src.dtype.type(hi - lo)).astype(src.dtype)
            out[i] = acc
        elif reduce == "min":
            out[i] = seg.min(axis=0)
        else:
            out[i] = seg.max(axis=0)
    return out


def segment_csr_fast(src: np.ndarray, indptr: np.ndarray, reduce: str = "sum") -> np.ndarray:
    """Vectorised segment_csr for large inputs (same results for integer
    dtypes; floating sums may differ from the sequential order in the last
    bits, so compare with a tolerance)."""
    src = np.asarray(src)
    indptr = np.asarray(indptr, dtype=np.int64)
    nseg = indptr.size - 1
    out = np.zeros((nseg,) + src.shape[1:], dtype=src.dtype)
    cnt = indptr[1:] - indptr[:-1]
    nz = cnt > 0
    if not nz.any() or src.shape[0] == 0:
        return out
    starts = indptr[:-1][nz]
    ufunc = {"sum": np.add, "add": np.add, "mean": np.add, "min": np.minimum, "max": np.maximum}[reduce]
    # reduceat reduces [starts[i], starts[i+1]); empty segments were dropped,
    # so consecutive non-empty starts delimit exactly one segment each except
    # the last, which must stop at its own end.
    last_end = int(indptr[1:][nz][-1])
    red = ufunc.reduceat(src[:last_end], starts, axis=0)
    if reduce == "mean":
        c = cnt[nz].reshape((-1,) + (1,) * (src.ndim - 1))
        red = red // c if _is_int(src) else (red / c.astype(src.dtype)).astype(src.dtype)
    out[nz] = red
    return out


def scatter(src: np.ndarray, index: np.ndarray, dim_size: int, reduce: str = "sum") -> np.ndarray:
    """pytorch_scatter.scatter(src, index, dim=0, dim_size=..., reduce) as used
    by reduce.py:42 and storage.py:414 (scatter_add)."""
    assert reduce in REDUCTIONS
    src = np.asarray(src)
    index = np.asarray(index, dtype=np.int64)
    out = np.zeros((dim_size,) + src.shape[1:], dtype=src.dtype)
    if reduce in ("sum", "add", "mean"):
        np.add.at(out, index, src)
        if reduce == "mean":
            cnt = np.bincount(index, minlength=dim_size).astype(np.int64)
            cnt = np.maximum(cnt, 1).reshape((-1,) + (1,) * (src.ndim - 1))
            out = out // cnt if _is_int(src) else (out / cnt.astype(src.dtype)).astype(src.dtype)
        return out
    touched = np.zeros(dim_size, dtype=bool)
    touched[index] = True
    if reduce == "min":
        init = np.iinfo(src.dtype).max if _is_int(src) else np.inf
        out[...] = init
        np.minimum.at(out, index, src)
    else:
        init = np.iinfo(src.dtype).min if _is_int(src) else -np.inf
        out[...] = init
        np.maximum.at(out, index, src)
    out[~touched] = 0
    return out


class Storage:
    """Plain-numpy mirror of SparseStorage's canonical state + lazy caches."""

    def __init__(self, row, col, value=None, sparse_sizes=None, is_sorted=False):
        row = np.asarray(row, dtype=np.int64)
        col = np.asarray(col, dtype=np.int64)
        # storage.py:65-91 size inference
        M = sparse_sizes[0] if sparse_sizes and sparse_sizes[0] is not None else (int(row.max()) + 1 if row.size else 0)
        N = sparse_sizes[1] if sparse_sizes and sparse_sizes[1] is not None else (int(col.max()) + 1 if col.size else 0)
        self.M, self.N = int(M), int(N)
        self.row, self.col = row, col
        self.value = None if value is None else np.asarray(value)
        # storage.py:158-171
        if not is_sorted and col.size > 0:
            key = row * self.N + col
            if (key[1:] < key[:-1]).any():
                perm = index_sort(key)
                self.row, self.col = row[perm], col[perm]
                if self.value is not None:
                    self.value = self.value[perm]

    # storage.py:211-222 via csrc/cpu/convert_cpu.cpp:6-30
    def rowptr(self):
        return np.searchsorted(self.row, np.arange(self.M + 1), side="left").astype(np.int64)

    def rowcount(self):  # storage.py:373-381
        p = self.rowptr()
        return p[1:] - p[:-1]

    def colcount(self):  # storage.py:405-420
        return np.bincount(self.col, minlength=self.N).astype(np.int64)

    def colptr(self):  # storage.py:386-400
        out = np.zeros(self.N + 1, dtype=np.int64)
        out[1:] = np.cumsum(self.colcount())
        return out

    def csr2csc(self):  # storage.py:425-434
        return index_sort(self.M * self.col + self.row)

    def csc2csr(self):  # storage.py:439-447
        return index_sort(self.csr2csc())

    def is_coalesced(self):  # storage.py:449-452
        key = np.concatenate([[-1], self.N * self.row + self.col])
        return bool((key[1:] > key[:-1]).all())

    def coalesce(self, reduce="add"):  # storage.py:454-486
        key = np.concatenate([[-1], self.N * self.row + self.col])
        mask = key[1:] > key[:-1]
        if mask.all():
            return self
        value = self.value
        if value is not None:
            ptr = np.concatenate([np.nonzero(mask)[0], [value.shape[0]]]).astype(np.int64)
            value = segment_csr_fast(value, ptr, reduce) if value.shape[0] > 4096 else segment_csr(value, ptr, reduce)
        return Storage(self.row[mask], self.col[mask], value, (self.M, self.N), is_sorted=True)


def coalesce(index, value, m, n, op="add"):
    """coalesce.py:25-29."""
    index = np.asarray(index, dtype=np.int64)
    st = Storage(index[0], index[1], value, (m, n), is_sorted=False).coalesce(op)
    return np.stack([st.row, st.col]), st.value


def transpose(index, value, m, n, coalesced=True):
    """transpose.py:41-65."""
    index = np.asarray(index, dtype=np.int64)
    row, col = index[1], index[0]
    if coalesced:
        st = Storage(row, col, value, (n, m), is_sorted=False).coalesce()
        row, col, value = st.row, st.col, st.value
    return np.stack([row, col]), value


def t(st: Storage) -> Storage:
    """transpose.py:9-33 (method form): permute by csr2csc, swap roles."""
    perm = st.csr2csc()
    value = None if st.value is None else st.value[perm]
    return Storage(st.col[perm], st.row[perm], value, (st.N, st.M), is_sorted=True)


def to_symmetric(st: Storage, reduce="sum") -> Storage:
    """tensor.py:415-451: entries of A and A^T together, duplicates reduced."""
    N = max(st.M, st.N)
    row = np.concatenate([st.row, st.col])
    col = np.concatenate([st.col, st.row])
    value = None if st.value is None else np.concatenate([st.value, st.value])
    return Storage(row, col, value, (N, N), is_sorted=False).coalesce(reduce)


def add(a: Storage, b: Storage) -> Storage:
    """add.py:30-47: concatenate both operands, coalesce with "sum"."""
    M, N = max(a.M, b.M), max(a.N, b.N)
    value = None
    if a.value is not None and b.value is not None:
        value = np.concatenate([a.value, b.value])
    return Storage(np.concatenate([a.row, b.row]), np.concatenate([a.col, b.col]), value,
                   (M, N), is_sorted=False).coalesce("sum")


def mul(a: Storage, b: Storage) -> Storage:
    """mul.py:35-73: entries present in BOTH (coalesced) operands, values multiplied."""
    M, N = max(a.M, b.M), max(a.N, b.N)
    key = np.concatenate([a.row * N + a.col, b.row * N + b.col])
    value = np.concatenate([a.value, b.value])
    perm = index_sort(key)
    key, value = key[perm], value[perm]
    hit = np.nonzero(key[1:] == key[:-1])[0]
    return Storage(key[hit] // N, key[hit] % N, value[hit] * value[hit + 1], (M, N), is_sorted=True)


def dense(st: Storage) -> np.ndarray:
    out = np.zeros((st.M, st.N) + (() if st.value is None else st.value.shape[1:]),
                   dtype=np.float64 if st.value is None else st.value.dtype)
    out[st.row, st.col] = 1 if st.value is None else st.value
    return out


def reduction(st: Storage, dim, reduce="sum", dtype=np.float32):
    """reduce.py:12-71 for dim in {None, 0, 1}."""
    value = st.value
    if dim is None:
        if value is not None:
            return {"sum": value.sum, "add": value.sum, "mean": value.mean, "min": value.min, "max": value.max}[reduce]()
        return dtype(st.col.size) if reduce in ("sum", "add") else dtype(1)
    if dim == 0:
        if value is not None:
            return scatter(value, st.col, st.N, reduce)
        return st.colcount().astype(dtype) if reduce in ("sum", "add") else np.ones(st.N, dtype)
    if dim == 1:
        if value is not None:
            return segment_csr_fast(value, st.rowptr(), reduce)
        return st.rowcount().astype(dtype) if reduce in ("sum", "add") else np.ones(st.M, dtype)
    raise ValueError(dim)


# ---- slicing ------------------------------------------------------------
# index_select.py:13-96, masked_select.py:12-90, narrow.py:11-100, restated op for op.
# paddle_scatter.gather_csr(src, indptr) repeats src[j] over segment j (pytorch_scatter semantics).
# Each function returns (Storage, caches): the caches the reference hands to the new
# SparseStorage (rowptr / rowcount / colptr / colcount / csc2csr), None where it passes None.

def _gather_csr(src: np.ndarray, indptr: np.ndarray) -> np.ndarray:
    return np.repeat(src, np.diff(indptr))


def index_select(st: Storage, dim: int, idx) -> tuple:
    """index_select.py:13-80."""
    idx = np.asarray(idx, dtype=np.int64)
    ndim = 2 + (0 if st.value is None else st.value.ndim - 1)
    dim = ndim + dim if dim < 0 else dim
    if dim == 0:  # :17-44
        old_rowptr = st.rowptr()
        rowcount = st.rowcount()[idx]
        rowptr = np.zeros(idx.size + 1, np.int64)
        rowptr[1:] = np.cumsum(rowcount)
        row = np.repeat(np.arange(idx.size, dtype=np.int64), rowcount)
        perm = np.arange(row.size, dtype=np.int64) + _gather_csr(old_rowptr[idx] - rowptr[:-1], rowptr)
        value = None if st.value is None else st.value[perm]
        return (Storage(row, st.col[perm], value, (idx.size, st.N), is_sorted=True),
                {"rowptr": rowptr, "rowcount": rowcount})
    if dim == 1:  # :46-75
        to_csc = st.csr2csc()
        old_colptr, row_csc = st.colptr(), st.row[to_csc]
        value_csc = None if st.value is None else st.value[to_csc]
        colcount = st.colcount()[idx]
        colptr = np.zeros(idx.size + 1, np.int64)
        colptr[1:] = np.cumsum(colcount)
        col = np.repeat(np.arange(idx.size, dtype=np.int64), colcount)
        perm = np.arange(col.size, dtype=np.int64) + _gather_csr(old_colptr[idx] - colptr[:-1], colptr)
        row = row_csc[perm]
        csc2csr = index_sort(idx.size * row + col)
        value = None if value_csc is None else value_csc[perm][csc2csr]
        return (Storage(row[csc2csr], col[csc2csr], value, (st.M, idx.size), is_sorted=True),
                {"colptr": colptr, "colcount": colcount, "csc2csr": csc2csr})
    if st.value is None:  # :76-80
        raise ValueError
    return Storage(st.row, st.col, np.take(st.value, idx, axis=dim - 1), (st.M, st.N), is_sorted=True), {}


def index_select_nnz(st: Storage, idx, layout=None) -> Storage:
    """index_select.py:83-101."""
    idx = np.asarray(idx, dtype=np.int64)
    if layout == "csc":
        idx = st.csc2csr()[idx]
    value = None if st.value is None else st.value[idx]
    return Storage(st.row[idx], st.col[idx], value, (st.M, st.N), is_sorted=True)


def masked_select(st: Storage, dim: int, mask) -> tuple:
    """masked_select.py:12-77."""
    mask = np.asarray(mask, dtype=bool)
    ndim = 2 + (0 if st.value is None else st.value.ndim - 1)
    dim = ndim + dim if dim < 0 else dim
    if dim == 0:  # :18-40
        rowcount = st.rowcount()[mask]
        emask = mask[st.row]
        row = np.repeat(np.arange(rowcount.size, dtype=np.int64), rowcount)
        value = None if st.value is None else st.value[emask]
        return Storage(row, st.col[emask], value, (rowcount.size, st.N), is_sorted=True), {"rowcount": rowcount}
    if dim == 1:  # :42-70
        to_csc = st.csr2csc()
        row, col = st.row[to_csc], st.col[to_csc]
        colcount = st.colcount()[mask]
        emask = mask[col]
        col = np.repeat(np.arange(colcount.size, dtype=np.int64), colcount)
        row = row[emask]
        csc2csr = index_sort(colcount.size * row + col)
        value = None if st.value is None else st.value[to_csc][emask][csc2csr]
        return (Storage(row[csc2csr], col[csc2csr], value, (st.M, colcount.size), is_sorted=True),
                {"colcount": colcount, "csc2csr": csc2csr})
    if st.value is None:  # :71-77
        raise ValueError
    return Storage(st.row, st.col, np.take(st.value, np.nonzero(mask)[0], axis=dim - 1), (st.M, st.N), is_sorted=True), {}


def masked_select_nnz(st: Storage, mask, layout=None) -> Storage:
    """masked_select.py:80-96."""
    mask = np.asarray(mask, dtype=bool)
    if layout == "csc":
        mask = mask[st.csc2csr()]
    value = None if st.value is None else st.value[mask]
    return Storage(st.row[mask], st.col[mask], value, (st.M, st.N), is_sorted=True)


def narrow(st: Storage, dim: int, start: int, length: int) -> tuple:
    """narrow.py:11-100."""
    ndim = 2 + (0 if st.value is None else st.value.ndim - 1)
    if dim < 0:
        dim = ndim + dim
    sizes = (st.M, st.N) + (() if st.value is None else st.value.shape[1:])
    if start < 0:
        start = sizes[dim] + start
    if dim == 0:  # :17-50
        rowptr = st.rowptr()[start:start + length + 1]
        row_start = rowptr[0]
        rowptr = rowptr - row_start
        row_length = rowptr[-1]
        sl = slice(row_start, row_start + row_length)
        value = None if st.value is None else st.value[sl]
        return (Storage(st.row[sl] - start, st.col[sl], value, (length, st.N), is_sorted=True),
                {"rowptr": rowptr, "rowcount": st.rowcount()[start:start + length]})
    if dim == 1:  # :52-87
        mask = (st.col >= start) & (st.col < start + length)
        value = None if st.value is None else st.value[mask]
        colptr = st.colptr()[start:start + length + 1]
        return (Storage(st.row[mask], st.col[mask] - start, value, (st.M, length), is_sorted=True),
                {"colptr": colptr - colptr[0], "colcount": st.colcount()[start:start + length]})
    if st.value is None:  # :88-96
        raise ValueError
    sl = [slice(None)] * st.value.ndim
    sl[dim - 1] = slice(start, start + length)
    return Storage(st.row, st.col, st.value[tuple(sl)], (st.M, st.N), is_sorted=True), {}
