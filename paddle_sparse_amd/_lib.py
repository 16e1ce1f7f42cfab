"""ctypes binding of the C-ABI HIP core (include/paddle_sparse_hip.h).

The library is the product: if it is missing or does not load, importing the
ops fails loudly — there is no CPU or eager fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_int64, c_size_t, c_uint64, c_void_p, c_char_p
from pathlib import Path

# The framework must bring its HIP runtime into the process FIRST: the core
# only has a NEEDED entry for libamdhip64.so.7 and binds to the copy that is
# already loaded.  Loaded the other way round, the process ends up with two
# HIP runtimes and the core's launches see "no ROCm-capable device".
import torch  # noqa: F401

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libpaddle_sparse_hip.so"

SUM, MEAN, MIN, MAX = 0, 1, 2, 3
REDUCE_ID = {"sum": SUM, "add": SUM, "mean": MEAN, "min": MIN, "max": MAX}
SPMM_ALGO_ID = {"auto": 0, "row_waves": 1, "edge_ranges": 2}

# name -> (restype, argtypes); mirrors include/paddle_sparse_hip.h one to one
# (tests/test_abi.py parses the header and checks this table against it).
SIGNATURES = {
    "psa_last_error": (c_char_p, []),
    "psa_abi_version": (c_int, []),
    "psa_sparse_cuda_version": (c_int64, []),
    "psa_ind2ptr": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "psa_ptr2ind": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "psa_spmm_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "psa_spmm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                         c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                         c_size_t, c_void_p]),
    "psa_spmm_coo": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                             c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p,
                             c_size_t, c_void_p]),
    "psa_spmm_half_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "psa_spmm_half_coo": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                  c_size_t, c_void_p]),
    "psa_spmm_half": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64,
                              c_int64, c_void_p, c_void_p, c_void_p]),
    "psa_spmm_half_sum_bw_csc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                         c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "psa_spmm_half_bw_csc_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "psa_spmm_half_arg": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64,
                                  c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "psa_spmm_half_minmax_bw_csc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_size_t, c_void_p]),
    "psa_spmm_half_set_variant": (c_int, [c_int]),
    "psa_csr_row_stats": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "psa_spmm_set_variant": (c_int, [c_int]),
    "psa_spmm_value_bw_workspace_bytes": (c_size_t, [c_int64]),
    "psa_spmm_value_bw": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "psa_transpose_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_int, c_void_p, c_void_p]),
    "psa_spmm_minmax_bw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "psa_csc_edge_tags": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "psa_spmm_minmax_bw_csc_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "psa_spmm_minmax_bw_csc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                       c_int64, c_int64, c_int64,
                                       c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "psa_spmm_minmax_bw_eb_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "psa_spmm_minmax_bw_eb": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "psa_spmm_sum_bw_csc_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "psa_spmm_sum_bw_csc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_int64,
                                    c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "psa_index_sort_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "psa_index_sort": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "psa_index_sort_status": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "psa_sort_pairs_u32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "psa_sort_pairs_u32_field": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "psa_merge_sorted": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "psa_sort_set_variant": (c_int, [c_int]),
    "psa_sort_set_spin_limit": (c_int64, [c_int64]),
    "psa_coalesce_small_max": (c_int64, []),
    "psa_coalesce_small_workspace_bytes": (c_size_t, [c_int64]),
    "psa_coalesce_small": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "psa_coalesce_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "psa_coalesce_count": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                                   c_void_p, c_size_t, c_void_p]),
    "psa_coalesce_write": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int64,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "psa_coalesce_small_max_fused": (c_int, []),
    "psa_coalesce_small_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "psa_make_keys_checked": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                      c_void_p]),
    "psa_make_keys": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                              c_void_p, c_void_p]),
    "psa_split_keys": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "psa_gather_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "psa_gather_rows_window": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "psa_invert_permutation": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "psa_permute_tile": (c_int64, []),
    "psa_permute_plan_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "psa_permute_apply_u32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "psa_bincount": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "psa_count2ptr_workspace_bytes": (c_size_t, [c_int64]),
    "psa_count2ptr": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "psa_unique_workspace_bytes": (c_size_t, [c_int64]),
    "psa_unique_count_after_sort": (c_int, [c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_void_p]),
    "psa_unique_count": (c_int, [c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "psa_unique_write": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "psa_unique_write_reduce": (c_int, [c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "psa_segment_reduce": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_int64, c_int64, c_void_p, c_void_p]),
    "psa_scatter_workspace_bytes": (c_size_t, [c_int64]),
    "psa_scatter_reduce": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int64, c_int64,
                                   c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "psa_sample_count": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "psa_sample_select": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                  c_int64, c_int, c_uint64, c_void_p, c_void_p]),
    "psa_relabel_mark": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                 c_void_p, c_void_p, c_void_p]),
    "psa_relabel_finish": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                   c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "psa_spspmm_count": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "psa_spspmm_expand": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                  c_void_p, c_void_p]),
}

_lib = None


class HipCoreError(RuntimeError):
    """A C-ABI call returned a non-zero psa_status."""


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: the HIP core is not built. Run "
            "`python -m paddle_sparse_amd.build` (needs hipcc, gfx950). "
            "paddle_sparse_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.psa_abi_version() != 1:
        raise ImportError("libpaddle_sparse_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load().psa_last_error().decode()
        raise HipCoreError(f"psa_status {status}: {msg}")
