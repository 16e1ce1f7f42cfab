"""TEST INFRASTRUCTURE — CPU oracle for the hot path.

Nothing under oracle/ is part of the product.  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import, load or
execute it, and only as the checker / the timed CPU baseline.

Contents
  convert_oracle.c   ind2ptr / ptr2ind, restating csrc/cpu/convert_cpu.cpp
  spmm_oracle.c      SpMM fwd/bwd (upstream pytorch_sparse algorithm; the
                     reference tree has no SpMM)
  storage_oracle.py  numpy restatement of storage.py / coalesce.py /
                     transpose.py / reduce.py / add.py / mul.py
  coalesce_oracle.c  C form of the sort + coalesce path (the timed CPU
                     baseline of those rows; checked against storage_oracle)
  spspmm_oracle.c    row-by-row sparse x sparse product (README.md:308-353)
  sample_oracle.c    sample_adj, restating csrc/cpu/sample_cpu.cpp

oracle/_ref (a build of the reference's own C++): NOT buildable in this
image — csrc/*.cpp include <paddle/extension.h> (no Paddle headers on disk)
and csrc/cpu/utils.h:4 includes parallel_hashmap/phmap.h from an empty,
un-fetched submodule.  The oracle is therefore pinned by the reference's
known-answer tests only (tests/golden/reference_kats.json).
"""
from __future__ import annotations

import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BUILD_DIR = HERE / "_build"
LIB_PATH = BUILD_DIR / "liboracle.so"
SOURCES = [HERE / "convert_oracle.c", HERE / "spmm_oracle.c", HERE / "coalesce_oracle.c",
           HERE / "spspmm_oracle.c", HERE / "sample_oracle.c"]

SUM, MEAN, MIN, MAX = 0, 1, 2, 3
REDUCE_ID = {"sum": SUM, "add": SUM, "mean": MEAN, "min": MIN, "max": MAX}


def build(force: bool = False) -> Path:
    """gcc recipe for the C restatement (separate roundings, OpenMP for the
    row-parallel baseline entry point)."""
    BUILD_DIR.mkdir(exist_ok=True)
    newest = max(s.stat().st_mtime for s in SOURCES)
    if not force and LIB_PATH.exists() and LIB_PATH.stat().st_mtime >= newest:
        return LIB_PATH
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found; cannot build the oracle")
    cmd = [gcc, "-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
           "-Wall", "-o", str(LIB_PATH), *map(str, SOURCES)]
    subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = ctypes.CDLL(str(LIB_PATH))
    return _lib


def _p(a, ctype):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctype))


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def ind2ptr(ind, M: int) -> np.ndarray:
    ind = _i64(ind)
    out = np.empty(M + 1, dtype=np.int64)
    lib().oracle_ind2ptr(_p(ind, ctypes.c_int64), ctypes.c_int64(ind.size),
                         ctypes.c_int64(M), _p(out, ctypes.c_int64))
    return out


def ptr2ind(ptr, E: int) -> np.ndarray:
    ptr = _i64(ptr)
    out = np.empty(E, dtype=np.int64)
    lib().oracle_ptr2ind(_p(ptr, ctypes.c_int64), ctypes.c_int64(ptr.size - 1),
                         ctypes.c_int64(E), _p(out, ctypes.c_int64))
    return out


def index_sort_c(keys, max_value: int, threads: int = 1):
    """(sorted keys, stable permutation) — coalesce_oracle.c."""
    keys = _i64(keys)
    out, perm = np.empty_like(keys), np.empty_like(keys)
    lib().oracle_index_sort(_p(keys, ctypes.c_int64), ctypes.c_int64(keys.size),
                            ctypes.c_int64(max_value), _p(out, ctypes.c_int64),
                            _p(perm, ctypes.c_int64), ctypes.c_int(threads))
    return out, perm


def coalesce_c(row, col, value, M: int, N: int, reduce: str = "add", threads: int = 1):
    """coalesce.py:25-29 on fp32 values [nnz] or [nnz, D] (or None) — coalesce_oracle.c.
    Returns (index[2, nnz'], value')."""
    row, col, value = _i64(row), _i64(col), _f32(value)
    nnz = row.size
    D = 0 if value is None else int(np.prod(value.shape[1:], dtype=np.int64))
    out_row, out_col = np.empty(nnz, np.int64), np.empty(nnz, np.int64)
    out_val = None if value is None else np.empty((nnz, D), np.float32)
    fn = lib().oracle_coalesce_f32
    fn.restype = ctypes.c_int64
    cnt = fn(_p(row, ctypes.c_int64), _p(col, ctypes.c_int64), _p(value, ctypes.c_float),
             ctypes.c_int64(D), ctypes.c_int64(nnz), ctypes.c_int64(M), ctypes.c_int64(N),
             ctypes.c_int(REDUCE_ID[reduce]), _p(out_row, ctypes.c_int64), _p(out_col, ctypes.c_int64),
             _p(out_val, ctypes.c_float), ctypes.c_int(threads))
    index = np.stack([out_row[:cnt], out_col[:cnt]])
    if value is None:
        return index, None
    return index, out_val[:cnt].reshape((cnt,) + value.shape[1:])


def spspmm(indexA, valueA, indexB, valueB, m: int, k: int, n: int):
    """README.md:308-353 on coalesced COO operands -> (index[2, nnz'], value') — spspmm_oracle.c."""
    indexA, indexB = _i64(indexA), _i64(indexB)
    valueA, valueB = _f32(valueA), _f32(valueB)
    rowptrA, rowptrB = ind2ptr(indexA[0], m), ind2ptr(indexB[0], k)
    colA, colB = _i64(indexA[1]), _i64(indexB[1])
    fn = lib().oracle_spspmm_f32
    fn.restype = ctypes.c_int64
    args = [_p(rowptrA, ctypes.c_int64), _p(colA, ctypes.c_int64), _p(valueA, ctypes.c_float),
            _p(rowptrB, ctypes.c_int64), _p(colB, ctypes.c_int64), _p(valueB, ctypes.c_float),
            ctypes.c_int64(m), ctypes.c_int64(n)]
    nnz = fn(*args, None, None, None)
    row, col = np.empty(nnz, np.int64), np.empty(nnz, np.int64)
    val = np.empty(nnz, np.float32)
    fn(*args, _p(row, ctypes.c_int64), _p(col, ctypes.c_int64), _p(val, ctypes.c_float))
    return np.stack([row, col]), val


def sample_adj(rowptr, col, idx, num_neighbors: int, replace: bool = False, seed: int = 0,
               num_nodes: int = None):
    """csrc/cpu/sample_cpu.cpp:9-148 -> (out_rowptr, out_col, n_id, e_id) — sample_oracle.c."""
    rowptr, col, idx = _i64(rowptr), _i64(col), _i64(idx)
    S = idx.size
    if num_nodes is None:
        num_nodes = max(rowptr.size - 1, int(col.max()) + 1 if col.size else 0)
    deg = rowptr[idx + 1] - rowptr[idx] if S else np.zeros(0, np.int64)
    cap = int(deg.sum()) if num_neighbors < 0 else S * max(int(num_neighbors), 0)
    out_rowptr = np.empty(S + 1, np.int64)
    out_col, e_id = np.empty(cap, np.int64), np.empty(cap, np.int64)
    n_id = np.empty(S + cap, np.int64)
    fn = lib().oracle_sample_adj
    fn.restype = ctypes.c_int64
    n = fn(_p(rowptr, ctypes.c_int64), _p(col, ctypes.c_int64), _p(idx, ctypes.c_int64),
           ctypes.c_int64(S), ctypes.c_int64(num_nodes), ctypes.c_int64(num_neighbors),
           ctypes.c_int(int(bool(replace))), ctypes.c_uint64(seed & (2**64 - 1)), ctypes.c_int64(cap),
           _p(out_rowptr, ctypes.c_int64), _p(out_col, ctypes.c_int64), _p(n_id, ctypes.c_int64),
           _p(e_id, ctypes.c_int64))
    assert n >= 0
    E = int(out_rowptr[-1])
    return out_rowptr, out_col[:E].copy(), n_id[:n].copy(), e_id[:E].copy()


def spmm(reduce: str, rowptr, col, value, mat, threads: int = 1):
    """Returns (out, arg_out or None).  threads > 1 uses the OpenMP entry."""
    rowptr, col = _i64(rowptr), _i64(col)
    value, mat = _f32(value), _f32(mat)
    M, K = rowptr.size - 1, mat.shape[1]
    rid = REDUCE_ID[reduce]
    out = np.empty((M, K), dtype=np.float32)
    arg = np.empty((M, K), dtype=np.int64) if rid in (MIN, MAX) else None
    fn = lib().oracle_spmm if threads <= 1 else lib().oracle_spmm_omp
    if threads > 1:  # libgomp (a dependency of liboracle.so) resolves through the same handle
        lib().omp_set_num_threads(ctypes.c_int(threads))
    fn(ctypes.c_int(rid), _p(rowptr, ctypes.c_int64), _p(col, ctypes.c_int64),
       _p(value, ctypes.c_float), _p(mat, ctypes.c_float), ctypes.c_int64(M),
       ctypes.c_int64(K), ctypes.c_int64(col.size), _p(out, ctypes.c_float),
       _p(arg, ctypes.c_int64))
    return out, arg


def spmm_abs_sum(rowptr, col, value, mat) -> np.ndarray:
    rowptr, col = _i64(rowptr), _i64(col)
    value, mat = _f32(value), _f32(mat)
    M, K = rowptr.size - 1, mat.shape[1]
    out = np.empty((M, K), dtype=np.float64)
    lib().oracle_spmm_abs_sum(_p(rowptr, ctypes.c_int64), _p(col, ctypes.c_int64),
                              _p(value, ctypes.c_float), _p(mat, ctypes.c_float),
                              ctypes.c_int64(M), ctypes.c_int64(K),
                              _p(out, ctypes.c_double))
    return out


def spmm_value_bw(reduce: str, row, rowptr, col, mat, grad) -> np.ndarray:
    row, rowptr, col = _i64(row), _i64(rowptr), _i64(col)
    mat, grad = _f32(mat), _f32(grad)
    out = np.empty(col.size, dtype=np.float32)
    lib().oracle_spmm_value_bw(ctypes.c_int(REDUCE_ID[reduce]), _p(row, ctypes.c_int64),
                               _p(rowptr, ctypes.c_int64), _p(col, ctypes.c_int64),
                               _p(mat, ctypes.c_float), _p(grad, ctypes.c_float),
                               ctypes.c_int64(col.size), ctypes.c_int64(mat.shape[1]),
                               _p(out, ctypes.c_float))
    return out


def spmm_mat_bw(reduce: str, row, rowptr, col, value, grad, N: int) -> np.ndarray:
    row, rowptr, col = _i64(row), _i64(rowptr), _i64(col)
    value, grad = _f32(value), _f32(grad)
    K = grad.shape[1]
    out = np.empty((N, K), dtype=np.float32)
    lib().oracle_spmm_mat_bw(ctypes.c_int(REDUCE_ID[reduce]), _p(row, ctypes.c_int64),
                             _p(rowptr, ctypes.c_int64), _p(col, ctypes.c_int64),
                             _p(value, ctypes.c_float), _p(grad, ctypes.c_float),
                             ctypes.c_int64(col.size), ctypes.c_int64(N),
                             ctypes.c_int64(K), _p(out, ctypes.c_float))
    return out


def spmm_minmax_bw(col, value, mat, grad, arg_out, want_value=True, want_mat=True):
    col, arg_out = _i64(col), _i64(arg_out)
    value, mat, grad = _f32(value), _f32(mat), _f32(grad)
    M, K = grad.shape
    N = mat.shape[0]
    gv = np.empty(col.size, dtype=np.float32) if want_value else None
    gm = np.empty((N, K), dtype=np.float32) if want_mat else None
    lib().oracle_spmm_minmax_bw(_p(col, ctypes.c_int64), _p(value, ctypes.c_float),
                                _p(mat, ctypes.c_float), _p(grad, ctypes.c_float),
                                _p(arg_out, ctypes.c_int64), ctypes.c_int64(M),
                                ctypes.c_int64(N), ctypes.c_int64(K),
                                ctypes.c_int64(col.size), _p(gv, ctypes.c_float),
                                _p(gm, ctypes.c_float))
    return gv, gm
