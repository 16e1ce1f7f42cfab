#!/usr/bin/env python3
"""max backward over the CSC view on R-MAT 21, N times (for rocprofv3): rmat_bw_prof.py <relabel 0|1> <hub copies 0|1>"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseStorage, ops  # noqa: E402

relabel, hot = bool(int(sys.argv[1])), bool(int(sys.argv[2]))
dev = torch.device("cuda", 0)
N, rowptr, row, col, val = rmat_graph(21, 20_000_000, dev, relabel=relabel)
B = torch.randn(N, 128, device=dev)
G = torch.randn(N, 128, device=dev)
st = SparseStorage(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
csr2csc, inv = st.csr2csc(), st.csc2csr()
out, _, words = ops._spmm("max", rowptr, col, val, B, want_arg_bytes=2, want_arg=False)
tags = st._csc_edge_tags(2)
plan = st._csc_view()._hot_columns()
rc = plan[1] if hot else st._row_in_csc_order()
ids = plan[0] if hot else None
for _ in range(12):
    ops.spmm_minmax_bw_csc(rowptr, st.colptr(), rc, csr2csc, tags, val, B, G, None, want_value=False, csc2csr=inv,
                           arg_bytes=words, hot_ids=ids)
torch.cuda.synchronize()
