#!/usr/bin/env python3
"""Backward passes over the CSC view at config-3 size, each timed alone (HIP events, mean of N calls):
the one-pass sum backward and the one-pass max backward as autograd runs them (bytes-only arg on this
graph), with and without the way of grad_value back to CSR order, plus autograd steps through the
tensor surface.  tools/bw_probe.py [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import event_ms, make_workload  # noqa: E402
from paddle_sparse_amd import SparseStorage, SparseTensor, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
M = N = 2_000_000
F = 128
nnz = 20_000_000
rowptr, col, val = make_workload(M, N, nnz, F, 2, dev)
g = torch.Generator(device=dev).manual_seed(1)
B = torch.randn(N, F, generator=g, device=dev)
G = torch.randn(M, F, generator=g, device=dev)
st = SparseStorage(rowptr=rowptr, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
csr2csc, colptr, row_csc, inv, tags = st.csr2csc(), st.colptr(), st._row_in_csc_order(), st.csc2csr(), st._csc_edge_tags()
out, _, arg_bytes = ops._spmm("max", rowptr, col, val, B, want_arg_bytes=1, want_arg=False)


def t(name, fn, nbytes=None):
    fn()
    ms = event_ms(fn, reps)
    extra = f"  {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s on {nbytes / 1e9:.2f} GB" if nbytes else ""
    print(f"{name:68s} {ms:7.3f} ms{extra}", flush=True)


bw = nnz * (8 + 8 + 4 + 4 * F + 4 + 16) + N * (8 + 8 * F)
t("sum backward, both gradients (kernel + grad_value to CSR order)", lambda: ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=inv), bw)
t("sum backward, grad_mat only through the same pass", lambda: ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, False), None)
t("max backward, both gradients (bytes only)", lambda: ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, tags, val, B, G, None, csc2csr=inv, arg_bytes=arg_bytes), bw + nnz * (F + 1))
t("max backward, grad_mat only", lambda: ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, tags, val, B, G, None, want_value=False, arg_bytes=arg_bytes), None)
gv = torch.randn(nnz, device=dev)
t("gather of a 4-byte array through csc2csr (grad_value's way back)", lambda: ops.gather_rows(gv, inv), nnz * 16)
plan = st._permute_plan("to_csr", force=True)
assert torch.equal(ops.permute_apply(gv, plan), ops.gather_rows(gv, inv))
t("  the same through the planned two-pass permutation", lambda: ops.permute_apply(gv, plan), nnz * 24)
t("sum backward, both gradients, planned way back", lambda: ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=inv, to_csr_plan=plan), bw)
t("max backward, both gradients, planned way back", lambda: ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, tags, val, B, G, None, csc2csr=inv, arg_bytes=arg_bytes, to_csr_plan=plan), bw + nnz * (F + 1))
to_csc = st._permute_plan("to_csc", force=True)
def sum_bw_streamed():
    v_csc = ops.permute_apply(val, to_csc)
    return ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=inv, to_csr_plan=plan, value_csc=v_csc)
def max_bw_streamed():
    v_csc = ops.permute_apply(val, to_csc)
    return ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, tags, val, B, G, None, csc2csr=inv, arg_bytes=arg_bytes, to_csr_plan=plan, value_csc=v_csc)
a0, b0 = ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=inv)
a1, b1 = sum_bw_streamed()
assert torch.equal(a0, a1) and torch.equal(b0, b1)
t("sum backward, both gradients, values permuted to CSC order first + planned way back", sum_bw_streamed, bw)
t("max backward, both gradients, values permuted to CSC order first + planned way back", max_bw_streamed, bw + nnz * (F + 1))
t("forward spmm_sum", lambda: ops.spmm_sum(rowptr, col, val, B), nnz * (12 + 4 * F) + M * (8 + 4 * F))

v = val.clone().requires_grad_()
Bt = B.clone().requires_grad_()
a = SparseTensor(rowptr=rowptr, col=col, value=v, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
a.storage.csr2csc(), a.storage.csc2csr(), a.storage._csc_edge_tags()
for reduce in ("sum", "mean", "max"):
    def step():
        v.grad = Bt.grad = None
        a.matmul(Bt, reduce).backward(G)
    for _ in range(3):  # the storage builds its planned routes on the second request: keep that out of the timed calls
        step()
    t(f"autograd step spmm_{reduce} fwd + bwd, trained values", step)
