"""Seeded synthetic inputs shared by the tests (SURVEY.md §8(d) generators)."""
import numpy as np


def random_csr(M, N, nnz, seed, with_value=True, sort_cols=False):
    """Uniform random CSR: rows sorted, duplicates allowed (legal in CSR)."""
    rng = np.random.default_rng(seed)
    row = np.sort(rng.integers(0, M, nnz, dtype=np.int64)) if M > 0 else np.zeros(0, np.int64)
    col = rng.integers(0, N, nnz, dtype=np.int64) if N > 0 else np.zeros(0, np.int64)
    if sort_cols and nnz:
        order = np.lexsort((col, row))
        row, col = row[order], col[order]
    rowptr = np.searchsorted(row, np.arange(M + 1), side="left").astype(np.int64)
    value = rng.standard_normal(nnz).astype(np.float32) if with_value else None
    return row, rowptr, col, value


def skewed_csr(M, N, seed, long_rows=(0,), long_deg=1000, base_deg=3):
    """A few very long rows among short ones, plus empty rows."""
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 2 * base_deg + 1, M)
    deg[rng.random(M) < 0.2] = 0
    for r in long_rows:
        deg[r % M] = long_deg
    rowptr = np.zeros(M + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    row = np.repeat(np.arange(M, dtype=np.int64), deg)
    col = rng.integers(0, N, nnz, dtype=np.int64)
    value = rng.standard_normal(nnz).astype(np.float32)
    return row, rowptr, col, value
