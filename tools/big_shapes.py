#!/usr/bin/env python3
"""Shapes past 2^31 ELEMENTS (not bytes) through the SpMM forward and both CSC
backward passes: 20M x 20M, 40M edges, K = 128 -> every dense operand has
2.56e9 elements.  Catches 32-bit index arithmetic that the test-suite sizes
cannot reach.  Checked against rocSPARSE (torch.sparse.mm) for the forward and
against the three-kernel / atomic forms for the backward passes."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from paddle_sparse_amd import SparseStorage, ops  # noqa: E402

M = N = 20_000_000
NNZ, K = 40_000_000, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(9)
key = torch.unique(torch.randint(0, M * N, (NNZ,), generator=g, device=dev))
row, col = torch.div(key, N, rounding_mode="floor"), key % N
del key
val = torch.randn(row.numel(), generator=g, device=dev)
B = torch.randn(N, K, generator=g, device=dev)
G = torch.randn(M, K, generator=g, device=dev)
st = SparseStorage(row=row, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
rowptr = st.rowptr()
print(f"{M} x {N}, nnz {row.numel()}, K {K}: dense operands hold {M * K:.3e} elements (> 2^31 = {2**31:.3e})")

out = ops.spmm_sum(rowptr, col, val, B)
ref = torch.sparse.mm(torch.sparse_csr_tensor(rowptr, col, val, size=(M, N)), B)
scale = torch.sparse.mm(torch.sparse_csr_tensor(rowptr, col, val.abs(), size=(M, N)), B.abs())
print("spmm_sum vs rocSPARSE: max err / sum|terms| =", float(((out - ref).abs() / (scale + 1e-30)).max()))
tail = slice(M - 1000, M)
print("  last rows non-trivial:", bool(out[tail].abs().sum() > 0), " first rows:", bool(out[:1000].abs().sum() > 0))
del ref, scale

csr2csc, colptr, row_csc, csc2csr = st.csr2csc(), st.colptr(), st._row_in_csc_order(), st.csc2csr()
gv1, gm1 = ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=csc2csr)
gv3 = ops.spmm_value_bw(None, rowptr, col, B, G, "sum")
gm3 = ops.spmm_sum(colptr, row_csc, ops.transpose_weights(val, csr2csc, None, None, False), G)
print("sum backward, one CSC pass vs three kernels: max |d grad_mat| =", float((gm1 - gm3).abs().max()),
      " max |d grad_value| =", float((gv1 - gv3).abs().max()))
del gm1, gm3, gv1, gv3

out, arg = ops.spmm_max(rowptr, col, val, B)
gv_c, gm_c = ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, st._csc_edge_tags(), val, B, G, arg,
                                    csc2csr=csc2csr)
gv_a, gm_a = ops.spmm_minmax_bw(col, val, B, G, arg)
print("max backward, one CSC pass vs atomics: max |d grad_mat| =", float((gm_c - gm_a).abs().max()),
      " max |d grad_value| =", float((gv_c - gv_a).abs().max()), " (|grad_mat| max", float(gm_a.abs().max()), ")")
torch.cuda.synchronize()
print("peak HBM in use: %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))

# ---- half-width operands of 5.1 GB each (>= 4 GiB: the 64-bit addressing instantiations run for real) ----
del out, arg, gv_c, gm_c, gv_a, gm_a
torch.cuda.empty_cache()
Bh, Gh = B.to(torch.bfloat16), G.to(torch.bfloat16)
print(f"bf16 operands: {Bh.numel() * 2 / 2**30:.2f} GiB each")
out_h = ops._spmm("sum", rowptr, col, val, Bh)[0]
ref_h = ops.spmm_sum(rowptr, col, val, Bh.float())
mag = ops.spmm_sum(rowptr, col, val.abs(), Bh.float().abs())
print("bf16 spmm_sum vs the fp32 kernel on the same rounded operand: max err / (2^-8 |ref| + 1e-5 sum|terms|) =",
      float(((out_h.float() - ref_h).abs() / (2.0 ** -8 * ref_h.abs() + 1e-5 * mag + 1e-30)).max()))
del ref_h, mag, out_h
w = ops.gather_rows(val, csr2csc)
gv_h, gm_h = ops.spmm_half_sum_bw_csc(colptr, row_csc, w, Bh, Gh, True)
gv_f, gm_f = ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, Bh.float(), Gh.float(), True, csc2csr=csc2csr)
gv_hc = ops.gather_rows(gv_h, csc2csr)
mag_m = ops.spmm_sum(colptr, row_csc, w.abs(), Gh.float().abs())
print("bf16 pass over the CSC view vs the fp32 pass on the rounded operands: grad_mat max err / (2^-8 |ref| + 1e-5 sum|terms|) =",
      float(((gm_h.float() - gm_f).abs() / (2.0 ** -8 * gm_f.abs() + 1e-5 * mag_m + 1e-30)).max()),
      " grad_value max |d| / max |ref| =", float((gv_hc - gv_f).abs().max() / gv_f.abs().max()))
del gm_h, gm_f, gv_h, gv_f, gv_hc, mag_m
out_m, _, words = ops._spmm("max", rowptr, col, val, Bh, want_arg=False, want_arg_bytes=1)
ref_m, arg_m = ops.spmm_max(rowptr, col, val, Bh.float())
deg = rowptr[1:] - rowptr[:-1]
local = (arg_m - rowptr[:-1, None]).clamp(max=255)
want = torch.where(arg_m < col.numel(), local, torch.full_like(local, 255)).to(torch.uint8)
print("bf16 spmm_max: out == rounded fp32 out:", bool(torch.equal(out_m, ref_m.to(torch.bfloat16))),
      " row-local winners == int64 arg_out's:", bool(torch.equal(words, want)), " (longest row", int(deg.max()), ")")
torch.cuda.synchronize()
print("peak HBM in use: %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
