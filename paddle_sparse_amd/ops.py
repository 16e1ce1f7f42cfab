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

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import REDUCE_ID, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


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
    with torch.cuda.device(ind.device):
        check(_lib.load().psa_ind2ptr(_ptr(ind), ind.numel(), M, _ptr(out), _stream()))
    return out


def ptr2ind(ptr: torch.Tensor, E: int) -> torch.Tensor:
    """csrc/convert.cpp:46-76.  ptr: int64[M+1] -> int64[E]."""
    ptr = _index(ptr, "ptr")
    if ptr.numel() < 1:
        raise ValueError("ptr must have at least one element")
    out = torch.empty(E, dtype=torch.int64, device=ptr.device)
    with torch.cuda.device(ptr.device):
        check(_lib.load().psa_ptr2ind(_ptr(ptr), ptr.numel() - 1, E, _ptr(out), _stream()))
    return out


def _spmm(reduce: str, rowptr: torch.Tensor, col: torch.Tensor,
          value: Optional[torch.Tensor], mat: torch.Tensor
          ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    rowptr = _index(rowptr, "rowptr")
    col = _index(col, "col")
    _gpu(mat, "mat")
    if mat.dtype != torch.float32:
        raise TypeError(f"spmm is fp32 in this build (mat is {mat.dtype})")
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
    out = torch.empty((M, K), dtype=torch.float32, device=mat.device)
    arg = None
    if rid in (_lib.MIN, _lib.MAX):
        arg = torch.empty((M, K), dtype=torch.int64, device=mat.device)
    with torch.cuda.device(mat.device):
        check(_lib.load().psa_spmm(rid, _ptr(rowptr), _ptr(col), _ptr(value), _ptr(mat),
                                   M, N, K, nnz, _ptr(out), _ptr(arg), _stream()))
    return out, arg


def spmm_sum(rowptr, col, value, mat) -> torch.Tensor:
    """out[i] = sum_e value[e] * mat[col[e]] (value None -> weights 1)."""
    return _spmm("sum", rowptr, col, value, mat)[0]


def spmm_mean(rowptr, col, value, mat) -> torch.Tensor:
    return _spmm("mean", rowptr, col, value, mat)[0]


def spmm_min(rowptr, col, value, mat) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (out, arg_out); arg_out == nnz marks an empty row."""
    return _spmm("min", rowptr, col, value, mat)


def spmm_max(rowptr, col, value, mat) -> Tuple[torch.Tensor, torch.Tensor]:
    return _spmm("max", rowptr, col, value, mat)


def spmm_set_variant(variant: int) -> int:
    """Bench/test hook: pick the SpMM kernel variant (0 = auto)."""
    return _lib.load().psa_spmm_set_variant(int(variant))
