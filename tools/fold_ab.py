#!/usr/bin/env python3
"""Timings that feel the cross-lane folds (csrc/lane_fold.h) and the step structure of the backward kernels: training
steps in fp32 / bf16, spmm_value_bw, the passes over the CSC view (fp32, bf16, masked bf16) and the forwards, on config 3
and on R-MAT 21.  Run once per build:  python tools/fold_ab.py <tag>.  Two builds inside ONE gpurun call (the second
made on the box, e.g. with PSA_EXTRA_HIPCC_FLAGS=-D... python -m paddle_sparse_amd.build --force, or with an alternative
source file copied over) give a same-box A/B: profiles/r04_fold_ab.txt, r04_fold_hybrid_ab.txt, r04_tail_ab.txt,
r04_spmm_dpp_ab.txt."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import event_ms, make_workload, rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)
tag = sys.argv[1] if len(sys.argv) > 1 else "build"
F = 128


def steps(name, row, rowptr, col, val, M, N):
    B = torch.randn(N, F, device=dev)
    G = torch.randn(M, F, device=dev)
    v = val.clone().requires_grad_()
    Bt = B.clone().requires_grad_()
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(M, N), is_sorted=True, trust_data=True)

    def step(reduce, dense, grad):
        v.grad = dense.grad = None
        a.matmul(dense, reduce).backward(grad)

    for reduce in ("sum", "max"):
        for _ in range(6):
            step(reduce, Bt, G)
        print(f"[{tag}] {name}: spmm_{reduce} fwd + bwd, trained values, fp32: {event_ms(lambda: step(reduce, Bt, G), 10):7.3f} ms", flush=True)
    Bb = B.to(torch.bfloat16).requires_grad_()
    Gb = G.to(torch.bfloat16)
    for _ in range(6):
        step("sum", Bb, Gb)
    print(f"[{tag}] {name}: spmm_sum fwd + bwd, trained values, bf16: {event_ms(lambda: step('sum', Bb, Gb), 10):7.3f} ms", flush=True)
    st = a.storage
    fn = lambda: ops.spmm_value_bw(None, rowptr, col, B, G, "sum")
    fn()
    print(f"[{tag}] {name}: spmm_value_bw alone: {event_ms(fn, 10):7.3f} ms", flush=True)
    with torch.no_grad():
        fw = lambda: a.matmul(B, "sum")
        fw()
        print(f"[{tag}] {name}: spmm_sum forward (surface): {event_ms(fw, 10):7.3f} ms", flush=True)
    plan = st._csc_view()._hot_columns()
    w = ops.permute_apply(val, st._permute_plan("to_csc", force=True))
    bw = lambda: ops.spmm_sum_bw_csc(st.colptr(), st._row_in_csc_order() if plan is None else plan[1], st.csr2csc(), val, B, G, True,
                                     csc2csr=st.csc2csr(), hot_ids=None if plan is None else plan[0],
                                     to_csr_plan=st._permute_plan("to_csr", force=True), value_csc=w)
    bw()
    print(f"[{tag}] {name}: fp32 sum pass over the CSC view (both gradients, planned routes): {event_ms(bw, 10):7.3f} ms", flush=True)
    hb = lambda: ops.spmm_half_sum_bw_csc(st.colptr(), st._row_in_csc_order(), w, Bb.detach(), Gb, True,
                                          long_columns=st._csc_view()._longest_row() > 128)
    hb()
    print(f"[{tag}] {name}: bf16 sum pass over the CSC view (both gradients): {event_ms(hb, 10):7.3f} ms", flush=True)
    if st._longest_row() <= 65_535 and st._spmm_algo() == "row_waves":  # the masked half-width pass (min / max), as autograd feeds it
        width = 2 if st._longest_row() > 128 else 1
        _, _, words = ops._spmm("max", rowptr, col, val, Bb.detach(), want_arg=False, want_arg_bytes=width)
        tags = st._csc_edge_tags(width)
        mb = lambda: ops.spmm_half_minmax_bw_csc(st.colptr(), st._row_in_csc_order(), tags, w, Bb.detach(), Gb, words,
                                                 long_columns=st._csc_view()._longest_row() > 128)
        mb()
        print(f"[{tag}] {name}: bf16 masked pass over the CSC view (max, both gradients): {event_ms(mb, 10):7.3f} ms", flush=True)
        fm = lambda: ops._spmm("max", rowptr, col, val, Bb.detach(), want_arg=False, want_arg_bytes=width)
        fm()
        print(f"[{tag}] {name}: bf16 max forward leaving the row-local arg_out: {event_ms(fm, 10):7.3f} ms", flush=True)
        def step_max():
            v.grad = Bb.grad = None
            a.matmul(Bb, "max").backward(Gb)
        for _ in range(6):
            step_max()
        print(f"[{tag}] {name}: spmm_max fwd + bwd, trained values, bf16: {event_ms(step_max, 10):7.3f} ms", flush=True)


M = N = 2_000_000
rowptr, col, val = make_workload(M, N, 20_000_000, F, 2, dev)
steps("config 3", ops.ptr2ind(rowptr, col.numel()), rowptr, col, val, M, N)
del rowptr, col, val
torch.cuda.empty_cache()
N, rowptr, row, col, val = rmat_graph(21, 20_000_000, dev)
steps("R-MAT 21", row, rowptr, col, val, N, N)
