#!/usr/bin/env python3
"""XCD mixing of the row blocks of the CSC-view backward kernels (MaskArgs.mix_xcds), off
(variant 27) and on: the backward passes alone on the uniform config-3 graph and on R-MAT 21."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, make_workload, rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseStorage, _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
F = 128


def passes(name, N, rowptr, row, col, val):
    M = rowptr.numel() - 1
    B = torch.randn(N, F, device=dev)
    G = torch.randn(M, F, device=dev)
    st = SparseStorage(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    csr2csc, inv = st.csr2csc(), st.csc2csr()
    width = 2 if st._longest_row() > 128 else 1
    out, _, words = ops._spmm("max", rowptr, col, val, B, want_arg_bytes=width, want_arg=False)
    tags = st._csc_edge_tags(width)
    plan = st._csc_view()._hot_columns()
    rc, ids = (plan[1], plan[0]) if plan is not None else (st._row_in_csc_order(), None)
    fns = {
        "max bwd (grad_mat)": lambda: ops.spmm_minmax_bw_csc(rowptr, st.colptr(), rc, csr2csc, tags, val, B, G, None, want_value=False,
                                                             csc2csr=inv, arg_bytes=words, hot_ids=ids),
        "max bwd (both)": lambda: ops.spmm_minmax_bw_csc(rowptr, st.colptr(), rc, csr2csc, tags, val, B, G, None, want_value=True,
                                                         csc2csr=inv, arg_bytes=words, hot_ids=ids),
        "sum bwd (both)": lambda: ops.spmm_sum_bw_csc(st.colptr(), rc, csr2csc, val, B, G, True, csc2csr=inv, hot_ids=ids),
    }
    for label, fn in fns.items():
        line = f"{name}: {label}:"
        for variant, tag in ((27, "XCD x = blocks x mod 8"), (0, "mixed")):
            prev = lib.psa_spmm_set_variant(variant)
            try:
                fn()
                line += f"  {tag} {event_ms(fn, 10):.3f} ms"
            finally:
                lib.psa_spmm_set_variant(prev)
        print(line, flush=True)


M = 2_000_000
rowptr, col, val = make_workload(M, M, 20_000_000, F, 0, dev)
passes("uniform C3", M, rowptr, ops.ptr2ind(rowptr, col.numel()), col, val)
del rowptr, col, val
for relabel in (False, True):
    N, rowptr, row, col, val = rmat_graph(21, 20_000_000, dev, relabel=relabel)
    passes("R-MAT 21, columns relabelled" if relabel else "R-MAT 21 as generated", N, rowptr, row, col, val)
