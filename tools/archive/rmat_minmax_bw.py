#!/usr/bin/env python3
"""min/max forward + one-pass backward on the R-MAT scale-21 graph of
tools/archive/spmm_rmat.py (power-law rows AND columns): what do rows beyond the
one-byte form's reach cost the backward?"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from paddle_sparse_amd import SparseTensor, coalesce, ops  # noqa: E402

dev = torch.device("cuda", 0)
scale, n, F = 21, 20_000_000, 128
N = 1 << scale
g = torch.Generator(device=dev).manual_seed(4)
row = torch.zeros(n, dtype=torch.int64, device=dev)
col = torch.zeros(n, dtype=torch.int64, device=dev)
for bit in range(scale):
    r = torch.rand(n, generator=g, device=dev)
    right = ((r >= 0.57) & (r < 0.76)) | (r >= 0.95)
    down = r >= 0.76
    row |= down.to(torch.int64) << bit
    col |= right.to(torch.int64) << bit
index, val = coalesce(torch.stack([row, col]), torch.randn(n, generator=g, device=dev), N, N)
A = SparseTensor(row=index[0].contiguous(), col=index[1].contiguous(), value=val, sparse_sizes=(N, N), is_sorted=True,
                 trust_data=True)
st = A.storage
deg = st.rowcount()
nnz = st.col().numel()
for cut in (128, 255):
    print(f"edges in rows of more than {cut} entries: {int(deg[deg > cut].sum())} of {nnz}")
B = torch.randn(N, F, device=dev)
G = torch.randn(N, F, device=dev)


def ms(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


out, arg, ab = ops._spmm("max", st.rowptr(), st.col(), val, B, want_arg_bytes=True)
args = (st.rowptr(), st.colptr(), st._row_in_csc_order(), st.csr2csc(), st._csc_edge_tags(), val, B, G, arg)
inv = st.csc2csr()
print(f"spmm_max forward (+ arg_out + bytes): {ms(lambda: ops._spmm('max', st.rowptr(), st.col(), val, B, want_arg_bytes=True)):.3f} ms")
print(f"one-pass backward, both gradients:    {ms(lambda: ops.spmm_minmax_bw_csc(*args, csc2csr=inv, arg_bytes=ab)):.3f} ms")
print(f"one-pass backward, grad_mat only:     {ms(lambda: ops.spmm_minmax_bw_csc(*args, want_value=False, arg_bytes=ab)):.3f} ms")
gv, gm = ops.spmm_minmax_bw_csc(*args, csc2csr=inv, arg_bytes=ab)
gv2, gm2 = ops.spmm_minmax_bw(st.col(), val, B, G, arg)
print("against the atomics backward: max |grad_mat diff| / max |grad_mat| =",
      float((gm - gm2).abs().max() / gm2.abs().max()), " grad_value equal:", bool(torch.equal(gv, gv2)))
print(f"atomics backward, both gradients:     {ms(lambda: ops.spmm_minmax_bw(st.col(), val, B, G, arg)):.3f} ms")

# the sum backward on the same graph: one pass over the CSC view against the three-kernel form
w = ops.transpose_weights(val, st.csr2csc(), None, None, False)
t3 = ms(lambda: (ops.spmm_value_bw(None, st.rowptr(), st.col(), B, G, "sum"),
                 ops.transpose_weights(val, st.csr2csc(), None, None, False),
                 ops.spmm_sum(st.colptr(), st._row_in_csc_order(), w, G)))
t1 = ms(lambda: ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order(), st.csr2csc(), val, B, G, True, csc2csr=inv))
print(f"sum backward, both gradients: one CSC pass {t1:.3f} ms; value_bw + weight gather + SpMM over CSC {t3:.3f} ms")
print(f"spmm_sum forward: {ms(lambda: ops.spmm_sum(st.rowptr(), st.col(), val, B)):.3f} ms")
