#!/usr/bin/env python3
"""Round-2 SpMM suite: forward on the uniform config-3 graph and on R-MAT scale 21 (as
generated, and with relabelled columns), both kernel families, with and without the compact
copy of the hub rows of B; sum and max (max WITH arg_out, as the autograd forward needs it
on graphs with rows above 128 entries, and `out` only).  HIP-event times, algorithmic bytes
per SURVEY.md 8(d)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import algorithmic_bytes, event_ms, make_workload, rmat_graph  # noqa: E402
from paddle_sparse_amd import SparseTensor, ops  # noqa: E402

dev = torch.device("cuda", 0)
F = 128


def report(name, ms, nnz, alg):
    if name.startswith("R-MAT"):
        # The no-reuse byte model says nothing on a graph where 73 % of the entries hit 65 536 hub rows (they are
        # served from caches / the compact copy): quote the counters instead — 8.4-9.6 GB per launch
        # (profiles/r02_pmc_hot_columns.txt: FETCH_SIZE 4.16 M KiB x 1.705-2 + WRITE_SIZE 1.09 M KiB).
        lo, hi = (8.4, 9.6) if "+arg_out" not in name else (8.4 + 2.15, 9.6 + 2.15)  # + the int64 arg_out written
        print(f"{name:58s} {ms:7.3f} ms  {nnz / ms / 1e6:6.2f} GEdges/s  (counter bytes {lo:.1f}-{hi:.1f} GB/launch -> "
              f"{lo / ms:4.2f}-{hi / ms:4.2f} TB/s = {lo / ms / 8 * 100:4.1f}-{hi / ms / 8 * 100:4.1f} % of 8 TB/s; "
              f"the no-reuse model does not apply)", flush=True)
        return
    print(f"{name:58s} {ms:7.3f} ms  {nnz / ms / 1e6:6.2f} GEdges/s  {alg / ms / 1e9:5.2f} TB/s algorithmic "
          f"({alg / ms / 1e9 / 8 * 100:4.1f} % of 8 TB/s)", flush=True)


graphs = []
M, nnz = 2_000_000, 20_000_000
rowptr, col, val = make_workload(M, M, nnz, F, 2, dev)
graphs.append(("uniform C3", M, rowptr, ops.ptr2ind(rowptr, nnz), col, val))
for relabel in (False, True):
    N, rp, row, c, v = rmat_graph(21, 20_000_000, dev, relabel=relabel)
    graphs.append(("R-MAT 21, columns relabelled" if relabel else "R-MAT 21 as generated", N, rp, row, c, v))

for name, M, rowptr, row, col, val in graphs:
    nnz = col.numel()
    B = torch.randn(M, F, device=dev)
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
    st = a.storage
    plan = st._hot_columns()
    e, t, big, longest = ops.csr_row_stats(rowptr)
    print(f"== {name}: {nnz} entries, {e} empty rows, {t} rows of 1-2, {big} rows above 128, longest {longest}; "
          f"chosen: {st._spmm_algo()}, hot copy: {0 if plan is None else plan[0].numel()} rows", flush=True)
    for algo in ("row_waves", "edge_ranges"):
        kw = dict(row=row, algo=algo)
        for red, arg in (("sum", False), ("mean", False), ("max", True)):
            alg = algorithmic_bytes(nnz, M, F, True, arg)
            ops._spmm(red, rowptr, col, val, B, **kw)
            report(f"{name} [{algo}] spmm_{red}" + (" (+arg_out)" if arg else ""),
                   event_ms(lambda: ops._spmm(red, rowptr, col, val, B, **kw), 20), nnz, alg)
        ops._spmm("max", rowptr, col, val, B, want_arg=False, **kw)
        report(f"{name} [{algo}] spmm_max, out only",
               event_ms(lambda: ops._spmm("max", rowptr, col, val, B, want_arg=False, **kw), 20), nnz,
               algorithmic_bytes(nnz, M, F, True, False))
    if plan is not None:
        hot, col_eff = plan

        def hot_call(red, want_arg=True):
            return ops._spmm(red, rowptr, col_eff, val, B, row=row, algo="edge_ranges", want_arg=want_arg,
                             hot_rows=ops.gather_rows(B, hot))

        for red, arg, want in (("sum", False, True), ("max", True, True), ("max", False, False)):
            hot_call(red, want)
            report(f"{name} [edge_ranges + hot copy, packed per call] spmm_{red}" + (" (+arg_out)" if arg else "")
                   + ("" if want or red == "sum" else ", out only"),
                   event_ms(lambda: hot_call(red, want), 20), nnz, algorithmic_bytes(nnz, M, F, True, arg))
    with torch.no_grad():
        a.matmul(B, "sum")
        report(f"{name} [SparseTensor.matmul, per-matrix choice] sum", event_ms(lambda: a.matmul(B, "sum"), 20), nnz,
               algorithmic_bytes(nnz, M, F, True, False))
    del B
