#!/usr/bin/env python3
"""Where the min/max backward over the CSC view loses time on R-MAT as generated: the pass
with and without edge values (the 4-byte gather through csr2csc), with and without the
hub-row copies, next to the same graph with relabelled columns."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseStorage, ops  # noqa: E402

dev = torch.device("cuda", 0)
F = 128
for relabel in (False, True):
    N, rowptr, row, col, val = rmat_graph(21, 20_000_000, dev, relabel=relabel)
    name = "columns relabelled" if relabel else "as generated"
    B = torch.randn(N, F, device=dev)
    G = torch.randn(N, F, device=dev)
    st = SparseStorage(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
    csr2csc, inv = st.csr2csc(), st.csc2csr()
    out, _, words = ops._spmm("max", rowptr, col, val, B, want_arg_bytes=2, want_arg=False)
    tags = st._csc_edge_tags(2)
    plan = st._csc_view()._hot_columns()
    for hot in (False, True):
        for v in (val, None):
            for want_value in (False, True):
                rc = plan[1] if hot else st._row_in_csc_order()
                ids = plan[0] if hot else None
                fn = lambda: ops.spmm_minmax_bw_csc(rowptr, st.colptr(), rc, csr2csc, tags, v, B, G, None,
                                                    want_value=want_value, csc2csr=inv, arg_bytes=words, hot_ids=ids)
                fn()
                print(f"{name}: max bwd, hub copies {'on ' if hot else 'off'}, values {'yes' if v is not None else 'no '}, "
                      f"grad_value {'yes' if want_value else 'no '}: {event_ms(fn, 10):.3f} ms", flush=True)
    w = ops.transpose_weights(val, csr2csc, st._row_in_csc_order(), rowptr, False)
    print(f"{name}: transpose_weights (value[csr2csc]) {event_ms(lambda: ops.transpose_weights(val, csr2csc, st._row_in_csc_order(), rowptr, False), 10):.3f} ms")
