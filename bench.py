#!/usr/bin/env python3
"""Headline benchmark: SpMM GEdges/s + achieved-HBM fraction.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--op spmm_sum] [--config c3|c4] [--scaling strong|weak]

N = 1 (default): BASELINE.json config 3 — random CSR, 2M x 2M, nnz = 20M,
dense F = 128 fp32, one `spmm_sum` forward per step, inputs resident in HBM.
N > 1 (launched by torch.distributed.run, one rank per GPU), a step = exchange
of B over RCCL (all of it by all-gather or by direct peer copies, or only the
rows the rank's columns touch by all_to_all_single) + the rank-local SpMM with
the block's per-matrix plan (paddle_sparse_amd.distributed.RowPartitionedSpMM):
  --scaling strong (default for c3 = BASELINE.json's metric "2M-node nnz=20M
      F=128, 1/2/4/8 GPU"): the ONE config-3 matrix, rows split by nnz over the
      ranks, the ONE B row-sharded;
  --scaling weak (default for --config c4): the same per-GPU rows/edges on every
      rank: rank r owns rows [r*2M, (r+1)*2M) of A (20M edges, columns over all
      N*2M nodes) and the matching block of B; c4 runs F = 256: at 8 GPUs
      BASELINE config 4 (16M x 16M, 160M entries).

Prints ONE JSON line on rank 0 (contract in the task brief): whole-job
GEdges/s, the roofline object of the dominant kernel (algorithmic bytes /
HIP-event kernel time vs 8 TB/s) and the CPU baseline (oracle C port timed on
this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# numpy / torch are imported by load_compute_modules() — a launcher process (self_launch below) never needs them
np = torch = None


def load_compute_modules() -> None:
    global np, torch
    import numpy
    import torch as torch_mod

    np, torch = numpy, torch_mod

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak

M_PER_GPU = 2_000_000
NNZ_PER_GPU = 20_000_000
FEAT = 128


def algorithmic_bytes(nnz: int, M: int, F: int, has_value: bool = True, arg_out: bool = False) -> int:
    """no-reuse gather model with the API's int64 indices."""
    per_edge = 8 + (4 if has_value else 0) + 4 * F
    per_row = 8 + 4 * F + (8 * F if arg_out else 0)
    return nnz * per_edge + M * per_row


def make_workload(M: int, N: int, nnz: int, F: int, seed: int, device):
    """SURVEY.md §8(d) generator: uniform random CSR, duplicates allowed."""
    from paddle_sparse_amd import ops

    g = torch.Generator(device=device).manual_seed(seed)
    row = torch.sort(torch.randint(0, M, (nnz,), generator=g, device=device))[0]
    col = torch.randint(0, N, (nnz,), generator=g, device=device)
    val = torch.randn(nnz, generator=g, device=device)
    rowptr = ops.ind2ptr(row, M)
    del row
    return rowptr, col, val


def host_threads() -> int:
    """Cores this process may really use: affinity, capped by the cgroup CPU
    quota (the GPU box gives a 1-GPU job a 16-CPU share of a larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(rowptr, col, val, B, budget_s: float = 15.0):
    """Oracle C port (OpenMP over rows) timed on this box's host cores on the same workload (all of config 3: one pass
    takes ~0.4 s on 16 threads; best of up to 10 passes inside the budget); also used to check the GPU."""
    import oracle

    threads = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    M = rowptr.numel() - 1
    sample_rows = min(M, 4_000_000)  # config 3 whole (2 M rows); a bound for larger shapes
    rp = rowptr[: sample_rows + 1].cpu().numpy()
    e = int(rp[-1])
    c = col[:e].cpu().numpy()
    v = val[:e].cpu().numpy()
    Bh = B.cpu().numpy()
    oracle.spmm("sum", rp[:1001], c, v, Bh, threads=threads)  # warm the pool
    best, reps, t_all = float("inf"), 0, time.perf_counter()
    out = None
    while reps < 10 and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        out, _ = oracle.spmm("sum", rp, c, v, Bh, threads=threads)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    info = {
        "value": round(e / best / 1e9, 4),
        "unit": "GEdges/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"all {sample_rows} rows" if sample_rows == M else f"first {sample_rows} rows") +
                  f" ({e} edges) of the same CSR x the full B, oracle_spmm_omp, best of {reps}",
    }
    return info, out, sample_rows


def event_ms(fn, reps: int) -> float:
    """Mean device time of fn() over reps launches (HIP events on the current stream)."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def rmat_graph(scale: int, n: int, device, seed: int = 4, relabel: bool = False):
    """R-MAT (0.57, 0.19, 0.19, 0.05) at the given scale, coalesced; optionally with the column
    ids relabelled by a random permutation (Graph500 relabels vertices: hubs then sit anywhere
    in memory instead of at ids with few bits set).  Returns (N, rowptr, row, col, value)."""
    from paddle_sparse_amd import coalesce, ops

    N = 1 << scale
    g = torch.Generator(device=device).manual_seed(seed)
    row = torch.zeros(n, dtype=torch.int64, device=device)
    col = torch.zeros(n, dtype=torch.int64, device=device)
    for bit in range(scale):
        r = torch.rand(n, generator=g, device=device)
        row |= (r >= 0.76).to(torch.int64) << bit
        col |= (((r >= 0.57) & (r < 0.76)) | (r >= 0.95)).to(torch.int64) << bit
    index, val = coalesce(torch.stack([row, col]), torch.randn(n, generator=g, device=device), N, N)
    row, col = index[0].contiguous(), index[1].contiguous()
    if relabel:
        col = torch.randperm(N, generator=g, device=device)[col].contiguous()
    return N, ops.ind2ptr(row, N), row, col, val


def breadth(rowptr, col, val, B, reps: int):
    """BASELINE config 3 is spmm_{sum,mean,max} forward AND backward: step times of the other
    ops of the path on the same workload, each with its algorithmic-byte fraction of the HBM
    peak.  Forward: SURVEY.md §8(d), nnz*(12+4F) + M*(8+4F) [+ M*F*8 arg_out].  Backward with
    trained values: the bytes of the ONE pass over the CSC view that produces both gradients
     — per entry the row id, the CSR edge id, the value through it, the gathered
    grad_out row, the grad_value element and its way back to CSR order (8+8+4+4F+4+16), per column
    colptr, its own row of the dense operand and the grad_mat row (8+8F); min/max add the
    row-local arg_out the forward leaves (M*F bytes) and its gather (F+1 bytes per entry).
    SURVEY §8(d)'s three-kernel model of the same gradients (`survey_model_gb`: SpMM over the
    transpose + SDDMM-shaped value pass, each with its own full gather) is kept beside it: the
    fused pass moves about half of that, which is why the step can beat the model's "peak"."""
    from paddle_sparse_amd import SparseTensor, ops

    M, F, nnz = rowptr.numel() - 1, B.shape[1], col.numel()
    N = B.shape[0]
    fwd = algorithmic_bytes(nnz, M, F, True, False)
    fwd_arg = algorithmic_bytes(nnz, M, F, True, True)
    bwd = nnz * (8 + 8 + 4 + 4 * F + 4 + 16) + N * (8 + 8 * F)
    bwd_minmax = bwd + nnz * (F + 1)
    survey_bwd = algorithmic_bytes(nnz, N, F, True, False) + nnz * 16 + nnz * (8 + 8 + 8 * F + 4)
    out = {}

    WARM_MS = 60.0  # every leg runs this long untimed first: the same warm-up for all (see below)

    def put(name, fn, nbytes, survey=None, n=reps, cold=False):
        """Times fn over n launches after WARM_MS of untimed launches.  The shader clock ramps for ~40 ms after
        the chip has idled (profiles/r03_warmup_probe.txt: the drift comes back after 2 s of idle, not after
        empty_cache or with fresh operands); the fp32 kernels do not feel it, the half-width one does (0.98 ->
        0.88 ms over its first 40 launches).  cold=True also records the first 10 launches after 1 s of idle."""
        rec = {}
        if cold:
            fn()
            torch.cuda.synchronize()
            time.sleep(1.0)
            rec["ms_first_10_launches_after_1s_idle"] = round(event_ms(fn, 10), 4)
        fn()
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < WARM_MS:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
        ms = event_ms(fn, n)
        rec.update({"ms": round(ms, 4), "gedges_per_s": round(nnz / ms / 1e6, 3),
                    "algorithmic_gb": round(nbytes / 1e9, 3), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)})
        if survey is not None:
            rec["survey_model_gb"] = round(survey / 1e9, 3)
        out[name] = rec

    for op, nb in (("spmm_mean", fwd), ("spmm_max", fwd_arg)):
        fn = getattr(ops, op)
        put(f"{op}_fwd", lambda: fn(rowptr, col, val, B), nb)
    # half-width dense operand (bf16 B and out, fp32 sums): 2 F of every (12 + 2 F) bytes per edge
    Bh = B.to(torch.bfloat16)
    half_bytes = nnz * (8 + 4 + 2 * F) + M * (8 + 2 * F)
    put("spmm_sum_bf16_fwd", lambda: ops._spmm("sum", rowptr, col, val, Bh), half_bytes, cold=True)
    del Bh
    row = ops.ptr2ind(rowptr, nnz)
    G = torch.randn(M, F, device=B.device)
    v = val.clone().requires_grad_()
    Bt = B.clone().requires_grad_()
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    a.storage.csr2csc(), a.storage.csc2csr(), a.storage._csc_edge_tags()  # one-off CSC view of the matrix (cached)

    def fwd_bwd(reduce):
        v.grad = Bt.grad = None
        a.matmul(Bt, reduce).backward(G)

    for reduce, nb, sv in (("sum", fwd + bwd, fwd + survey_bwd), ("max", fwd + M * F + bwd_minmax, fwd_arg + survey_bwd)):
        put(f"spmm_{reduce}_fwd_bwd", lambda: fwd_bwd(reduce), nb, sv, n=max(3, reps // 4))
    # trained values with a half-width dense operand: half-width forward, ONE half-width pass over the CSC view for
    # both gradients (psa_spmm_half_sum_bw_csc), value[csr2csc] and grad_value's way back along planned routes
    vb = val.clone().requires_grad_()
    Bb = B.detach().to(torch.bfloat16, copy=True).requires_grad_()
    Gb = G.to(torch.bfloat16)
    ab = SparseTensor(row=row, rowptr=rowptr, col=col, value=vb, sparse_sizes=(M, N), is_sorted=True, trust_data=True)

    def half_step():
        vb.grad = Bb.grad = None
        ab.matmul(Bb, "sum").backward(Gb)

    half_bwd = nnz * (8 + 4 + 16 + 2 * F + 4 + 16) + N * (8 + 4 * F)
    put("spmm_sum_bf16_fwd_bwd", half_step, half_bytes + half_bwd, n=max(3, reps // 4))
    del ab, vb, Bb, Gb
    # fixed adjacency (gradient wrt the dense operand only): the backward is a forward over the CSC view
    del a, v, Bt
    fixed = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    fixed.storage.csr2csc()
    for dtype, name, nb in ((torch.float32, "spmm_sum_fwd_bwd_fixed_adjacency", fwd + algorithmic_bytes(nnz, N, F, True, False)),
                            (torch.bfloat16, "spmm_sum_bf16_fwd_bwd_fixed_adjacency", 2 * half_bytes)):
        Bd = B.detach().to(dtype, copy=True).requires_grad_()  # a leaf of its own (B itself stays as it is)
        Gd = G.to(dtype)

        def step():
            Bd.grad = None
            fixed.matmul(Bd, "sum").backward(Gd)

        put(name, step, nb, n=max(3, reps // 4))
        del Bd, Gd
    return out


def config2(device):
    """BASELINE config 2: spmm_sum forward, CSR 100k x 100k, nnz 1 M, F = 64 — a 45 us problem, timed over 200 calls
    through the tensor surface (which knows the longest row and skips the long-row launches) and as the raw op."""
    from paddle_sparse_amd import SparseTensor, ops

    M = N = 100_000
    nnz, F = 1_000_000, 64
    rowptr, col, val = make_workload(M, N, nnz, F, 1, device)
    B = torch.randn(N, F, device=device)
    a = SparseTensor(rowptr=rowptr, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    nb = algorithmic_bytes(nnz, M, F)
    out = {}
    with torch.no_grad():
        for name, fn in (("tensor_surface", lambda: a.matmul(B)), ("raw_op", lambda: ops.spmm_sum(rowptr, col, val, B))):
            for _ in range(50):
                fn()
            ms = event_ms(fn, 200)
            out[name] = {"us": round(ms * 1e3, 2), "gedges_per_s": round(nnz / ms / 1e6, 3),
                         "frac_of_hbm_peak": round(nb / ms / 1e6 / HBM_PEAK_GBS, 4)}
    out["algorithmic_gb"] = round(nb / 1e9, 4)
    return out


def power_law(device, F: int, reps: int):
    """The forward on a power-law graph (R-MAT scale 21, 19.5 M entries), both kernel families,
    as generated and with relabelled columns."""
    from paddle_sparse_amd import ops

    res = {}
    for relabel in (False, True):
        N, rowptr, row, col, val = rmat_graph(21, 20_000_000, device, relabel=relabel)
        B = torch.randn(N, F, device=device)
        nnz = col.numel()
        entry = {"nnz": nnz, "rows": N}
        for algo in ("row_waves", "edge_ranges"):
            for op in ("spmm_sum", "spmm_max"):
                fn = getattr(ops, op)
                fn(rowptr, col, val, B, row=row, algo=algo)
                ms = event_ms(lambda: fn(rowptr, col, val, B, row=row, algo=algo), reps)
                entry[f"{op}_{algo}_ms"] = round(ms, 4)
        # the tensor surface: per-matrix choice of the kernel family and, for hub columns, the
        # compact copy of the hot rows of B (packed inside every call)
        from paddle_sparse_amd import SparseTensor

        a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
        for reduce in ("sum", "max"):
            with torch.no_grad():
                a.matmul(B, reduce)
                entry[f"spmm_{reduce}_tensor_surface_ms"] = round(event_ms(lambda: a.matmul(B, reduce), reps), 4)
        if not relabel:  # a training step on the graph as generated: trained edge values, both gradients
            v = val.clone().requires_grad_()
            Bt = B.clone().requires_grad_()
            G = torch.randn(N, F, device=device)
            t = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(N, N), is_sorted=True, trust_data=True)

            def step(reduce):
                v.grad = Bt.grad = None
                t.matmul(Bt, reduce).backward(G)

            for reduce in ("sum", "max"):
                for _ in range(3):  # the storage builds its planned routes on the SECOND request: keep that out of the timed steps
                    step(reduce)
                entry[f"spmm_{reduce}_fwd_bwd_trained_values_ms"] = round(event_ms(lambda: step(reduce), max(3, reps // 4)), 4)
            # the max forward as autograd runs it (out + the two-byte row-local arg_out), against out only above
            entry["spmm_max_forward_under_autograd_ms"] = round(event_ms(lambda: t.matmul(Bt, "max"), reps), 4)
            del t, v
            # the same step with a bf16 dense operand: half-width forward (edge ranges), ONE half-width pass over the CSC
            # view for both gradients with the hub rows' long columns in chunks — no fp32 copies of B / grad_out
            vb = val.clone().requires_grad_()
            Bb = B.to(torch.bfloat16).requires_grad_()
            Gb = G.to(torch.bfloat16)
            tb = SparseTensor(row=row, rowptr=rowptr, col=col, value=vb, sparse_sizes=(N, N), is_sorted=True, trust_data=True)

            def half_step(reduce="sum"):
                vb.grad = Bb.grad = None
                tb.matmul(Bb, reduce).backward(Gb)

            for reduce in ("sum", "max"):  # max: the edge-range forward leaves the two-byte row-local arg_out in half width too
                for _ in range(3):
                    half_step(reduce)
                entry[f"spmm_{reduce}_bf16_fwd_bwd_trained_values_ms"] = round(event_ms(lambda: half_step(reduce), max(3, reps // 4)), 4)
            del tb, vb, Bb, Gb
            # ... and with a fixed adjacency (gradient wrt the dense operand only)
            fixed = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)

            def fixed_step(reduce):
                Bt.grad = None
                fixed.matmul(Bt, reduce).backward(G)

            for reduce in ("sum", "max"):
                fixed_step(reduce)
                entry[f"spmm_{reduce}_fwd_bwd_fixed_adjacency_ms"] = round(event_ms(lambda: fixed_step(reduce), max(3, reps // 4)), 4)
            del fixed, Bt, G
        entry["algo_chosen_by_row_stats"] = a.storage._spmm_algo()
        entry["hot_column_copy_rows"] = 0 if a.storage._hot_columns() is None else int(a.storage._hot_columns()[0].numel())
        res["rmat21_relabelled_columns" if relabel else "rmat21_as_generated"] = entry
        del B
    res["rmat24_as_generated"] = power_law_rmat24(device, F, max(3, reps // 2))
    return res


def power_law_rmat24(device, F: int, reps: int):
    """R-MAT scale 24, 100 M generated entries (the graph of BASELINE config 5): hub rows of ~140 k entries, beyond the
    65 535 the two-byte row-local arg_out names — the min / max training step still keeps no int64 arg_out (the rows
    concerned are reduced once more in pieces, matmul._huge_piece_winners)."""
    from paddle_sparse_amd import SparseTensor

    N, rowptr, row, col, val = rmat_graph(24, 100_000_000, device)
    nnz = col.numel()
    B = torch.randn(N, F, device=device)
    G = torch.randn(N, F, device=device)
    a = SparseTensor(row=row, rowptr=rowptr, col=col, value=val, sparse_sizes=(N, N), is_sorted=True, trust_data=True)
    entry = {"nnz": nnz, "rows": N, "longest_row": a.storage._longest_row()}
    hr = a.storage._huge_rows()
    entry["rows_above_65535_entries"] = 0 if hr is None else int(hr["rows"].numel())
    entry["entries_in_those_rows"] = 0 if hr is None else int(hr["ids"].numel())
    with torch.no_grad():
        for reduce in ("sum", "max"):
            a.matmul(B, reduce)
            entry[f"spmm_{reduce}_tensor_surface_ms"] = round(event_ms(lambda: a.matmul(B, reduce), reps), 4)
    v = val.clone().requires_grad_()
    Bt = B.clone().requires_grad_()
    t = SparseTensor(row=row, rowptr=rowptr, col=col, value=v, sparse_sizes=(N, N), is_sorted=True, trust_data=True)

    def step(reduce):
        v.grad = Bt.grad = None
        t.matmul(Bt, reduce).backward(G)

    import paddle_sparse_amd.matmul  # noqa: F401  (the module; the package exports a function of the same name)
    mm_mod = sys.modules["paddle_sparse_amd.matmul"]
    for reduce in ("sum", "max"):
        for _ in range(3):
            step(reduce)
        entry[f"spmm_{reduce}_fwd_bwd_trained_values_ms"] = round(event_ms(lambda: step(reduce), reps), 4)
    entry["spmm_max_forward_under_autograd_ms"] = round(event_ms(lambda: t.matmul(Bt, "max"), reps), 4)  # incl. the pieces
    mm_mod.HUGE_ROW_PIECES = False  # the route of round 3: int64 arg_out beside the one-byte form
    try:
        step("max")
        entry["spmm_max_fwd_bwd_trained_values_int64_arg_out_ms"] = round(event_ms(lambda: step("max"), reps), 4)
    finally:
        mm_mod.HUGE_ROW_PIECES = True
    entry["algo_chosen_by_row_stats"] = a.storage._spmm_algo()
    return entry


def visible_gpus() -> int:
    """GPUs this process could use, counted WITHOUT initialising the HIP runtime (torch.cuda.device_count() reads the
    driver's device list on this image; nothing here creates a context — the launcher must stay exec- and fork-safe)."""
    import torch as t

    return t.cuda.device_count()


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` typed as it stands (no launcher, WORLD_SIZE unset): start ONE fresh child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <the same flags>` — one rank per GPU —
    pass rank 0's single JSON line through to stdout and everything else to stderr, and return the child's exit code.
    The parent never touches the GPU, never imports paddle_sparse_amd and never exec()s; a child that fails or runs past
    PSA_BENCH_LAUNCH_TIMEOUT seconds (default 3000) makes the parent exit non-zero.  No retry."""
    import signal
    import socket
    import subprocess

    rehearse = os.environ.get("PSA_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    have = visible_gpus()
    need = 1 if rehearse else args.gpus
    if have < need:
        print(f"bench.py: --gpus {args.gpus} needs {need} visible GPU(s), this node shows {have}", file=sys.stderr)
        return 2
    with socket.socket() as s_:  # a free rendezvous port on the loopback (the container hostname may not resolve)
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    limit = float(os.environ.get("PSA_BENCH_LAUNCH_TIMEOUT", "3000"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, _ = child.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {args.gpus}-rank child ran past {limit:.0f} s; stopping its process group", file=sys.stderr)
        os.killpg(child.pid, signal.SIGTERM)  # exactly the group this call started (start_new_session)
        try:
            child.communicate(timeout=30)
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, signal.SIGKILL)
            child.communicate()
        return 124
    lines, rest = [], []
    for ln in stdout.splitlines():
        try:
            ok = isinstance(json.loads(ln), dict) and "metric" in json.loads(ln)
        except ValueError:
            ok = False
        (lines if ok else rest).append(ln)
    if rest:
        print("\n".join(rest), file=sys.stderr)
    if child.returncode != 0:
        print(f"bench.py: the {args.gpus}-rank child exited with {child.returncode}", file=sys.stderr)
        return child.returncode
    if len(lines) != 1:
        print(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 3
    print(lines[0], flush=True)
    return 0


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--op", default="spmm_sum", choices=["spmm_sum", "spmm_mean", "spmm_max", "spmm_min"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--config", default="c3", choices=["c3", "c4"],
                    help="c3 (default): the 2M x 2M, nnz = 20M, F = 128 matrix of BASELINE config 3; "
                         "c4: F = 256, 2M rows / 20M entries per GPU — at 8 GPUs BASELINE config 4 (16M x 16M, 160M entries)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="N > 1.  strong: the ONE config-3 matrix split by nnz over the N ranks (BASELINE.json's metric: "
                         "'2M-node nnz=20M F=128, 1/2/4/8 GPU'); weak: 2M rows / 20M entries PER GPU, columns over all "
                         "N x 2M nodes (config 4's structure).  auto = strong for c3, weak for c4")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the breadth / power-law legs after the timed region")
    ap.add_argument("--exchange", default="auto", choices=["auto", "full", "full_p2p", "halo"],
                    help="N > 1: all of B by all-gather, the same by direct peer copies, or only the rows this rank's "
                         "columns touch (all_to_all_single); auto times full and halo before the warmup "
                         "(full_p2p only by name)")
    ap.add_argument("--feature-chunks", type=int, default=0,
                    help="N > 1: exchange B in this many column slices and run the SpMM of a slice under the "
                         "exchange of the next ones (1: one exchange, then the SpMM; default 0: time 1 and 4 "
                         "before the warmup and keep the faster)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="dense operand and output: f32 (the metric's dtype, default) or bf16 (fp32 sums; half the bytes of B "
                         "per gathered row and, at N > 1, on the fabric) — an optional run, never the headline")
    ap.add_argument("--no-plan", action="store_true", help="N > 1: rank-local kernel on the raw arrays (algo auto) "
                                                           "instead of the block's per-matrix plan")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "LOCAL_RANK" not in os.environ and os.environ.get("WORLD_SIZE", "1") != str(args.gpus):
        # typed without a launcher (torch.distributed.run gives every rank LOCAL_RANK; a stray WORLD_SIZE=1 in the
        # environment is not one): become the launcher, before anything initialises the GPU in this process
        raise SystemExit(self_launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    load_compute_modules()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # Rehearsal hook (functional only): PSA_BENCH_REHEARSE_ON_ONE_GPU=1 lets N ranks SHARE cuda:0 and talk over gloo
    # (which moves device tensors) — the N > 1 code path of this file on a one-GPU box.  Its times say nothing.
    rehearse = os.environ.get("PSA_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from paddle_sparse_amd import ops

    # world == 1 runs without a process group unless a rehearsal asks for one
    use_dist = world > 1 or os.environ.get("PSA_BENCH_FORCE_DIST") == "1"
    dist = None
    stdout_fd = None
    if use_dist:
        import torch.distributed as dist  # noqa: PLC0415

        # RCCL writes a version banner to STDOUT when its communicator comes up;
        # the contract is ONE JSON line there, so fd 1 points at stderr until
        # the line is printed.
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    global M_PER_GPU, NNZ_PER_GPU
    if rehearse:  # gloo stages device tensors through the host (~1 GB/s): a tenth of the problem keeps the run in minutes
        M_PER_GPU, NNZ_PER_GPU = M_PER_GPU // 10, NNZ_PER_GPU // 10
    scaling = args.scaling if args.scaling != "auto" else ("strong" if args.config == "c3" else "weak")
    F = 256 if args.config == "c4" else FEAT
    reduce = args.op.split("_", 1)[1]
    ops.spmm_set_variant(args.variant)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def wall(fn, n):
        """Seconds per call of fn over n calls, barrier + synchronize on both sides."""
        sync_all()
        t_w = time.perf_counter()
        for _ in range(n):
            fn()
        sync_all()
        return (time.perf_counter() - t_w) / n

    # ---- the workload ------------------------------------------------------------------------
    if use_dist and scaling == "strong":
        # BASELINE.json's metric at N > 1: the ONE config-3 matrix (every rank generates it from the same seed
        # and keeps its block of rows, balanced by nnz) and the ONE dense B, row-sharded in equal blocks
        from paddle_sparse_amd.distributed import RowPartitionedSpMM, partition_rows_by_nnz, shard_csr

        M_glob = N = M_PER_GPU
        nnz_glob = NNZ_PER_GPU
        g_rowptr, g_col, g_val = make_workload(M_glob, N, nnz_glob, F, seed=2, device=device)
        bounds = partition_rows_by_nnz(g_rowptr, world)
        shard = shard_csr(g_rowptr, g_col, g_val, N, bounds, rank)
        del g_rowptr, g_col, g_val
        rowptr, col, val = shard.rowptr, shard.col, shard.value
        M, nnz = shard.num_rows, shard.nnz
        g = torch.Generator(device=device).manual_seed(100)
        B_glob = torch.randn(N, F, generator=g, device=device)
        nb = (N + world - 1) // world
        B_local = torch.zeros(nb, F, device=device)
        mine = B_glob[rank * nb:(rank + 1) * nb]
        B_local[:mine.shape[0]] = mine
        del B_glob, mine
        total_nnz = nnz_glob
    else:
        M, nnz = M_PER_GPU, NNZ_PER_GPU
        N = M * world
        rowptr, col, val = make_workload(M, N, nnz, F, seed=2 + rank, device=device)
        g = torch.Generator(device=device).manual_seed(100 + rank)
        B_local = torch.randn(M, F, generator=g, device=device)
        total_nnz = world * nnz
        if use_dist:
            from paddle_sparse_amd.distributed import RowPartitionedSpMM, RowShard

            shard = RowShard(rowptr, col, val, rank * M, (rank + 1) * M, N)
    half = args.dtype == "bf16"
    if half:
        B_local = B_local.to(torch.bfloat16)
    torch.cuda.empty_cache()

    forms = {}  # (exchange, chunks) -> RowPartitionedSpMM
    chosen = ("full", 1)
    tune_ms = {}
    TUNE_STEPS = 10
    if use_dist:
        # full_p2p (batch_isend_irecv) has never run on more than one GPU: it is timed only when asked for by name,
        # so that a first multi-GPU run cannot lose its headline to it
        exchanges = ["full", "halo"] if args.exchange == "auto" else [args.exchange]
        chunk_opts = [1, 4] if args.feature_chunks == 0 and F % 16 == 0 else [max(args.feature_chunks, 1)]
        # exchange / send buffers and the output live on the objects: a step allocates nothing (distributed.py)
        ops_by_exchange = {e: RowPartitionedSpMM(shard, reduce=reduce, exchange=e, plan=not args.no_plan, keep_output=True)
                           for e in exchanges}
        forms = {(e, c): ops_by_exchange[e] for e in exchanges for c in chunk_opts}
        # Which form is faster depends on the fabric and on the graph: before the warmup every form runs 3 untimed
        # steps (buffers, plans, communicator channels) and then TUNE_STEPS timed ones in steady state — the host far
        # ahead of the GPU, as in the timed region —; the slowest rank's time decides and every rank keeps the same form.
        spent = []
        for (e, c), o in forms.items():
            for _ in range(3):
                o(B_local, feature_chunks=c)
            spent.append(wall(lambda: o(B_local, feature_chunks=c), TUNE_STEPS))
            if rank == 0:
                print(f"[bench] tuned {e}/chunks{c}: {spent[-1] * 1e3:.3f} ms per step", file=sys.stderr, flush=True)
        spent_t = torch.tensor(spent, dtype=torch.float64, device=device)
        dist.all_reduce(spent_t, op=dist.ReduceOp.MAX)
        spent = [float(x) for x in spent_t]
        keys = list(forms)
        chosen = keys[spent.index(min(spent))]
        tune_ms = {f"{e}/chunks{c}": round(x * 1e3, 4) for (e, c), x in zip(keys, spent)}
        op = forms[chosen]
        step = lambda: op(B_local, feature_chunks=chosen[1])  # noqa: E731  exchange of B + local HIP SpMM
        B_full = op.exchange_only(B_local, chosen[1])  # the chosen form's assembled operand (the object's buffers)
    else:
        fn = getattr(ops, args.op)
        B_full = B_local
        step = lambda: fn(rowptr, col, val, B_full)  # noqa: E731

    def local_kernel():
        if use_dist:
            return op.spmm_only(B_full)
        return ops._spmm(reduce, rowptr, col, val, B_full)[0]

    for _ in range(args.warmup):
        step()
    sync_all()

    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if isinstance(out, tuple):
        out = out[0]

    # dominant kernel alone (HIP events on its launch stream), after the timed region
    kern_ms = event_ms(local_kernel, args.steps)

    # SURVEY.md §8(e) asks for the local-kernel aggregate, end-to-end serial and end-to-end
    # overlapped: the forms that were NOT timed above run a few steps here, so the line carries all.
    form_s = {}
    exch_ms = {}
    kern_by_exchange = {}
    if use_dist:
        other_steps = max(TUNE_STEPS, args.steps // 2)
        for key, o in forms.items():
            if key == chosen:
                form_s[key] = elapsed / args.steps
                continue
            for _ in range(2):
                o(B_local, feature_chunks=key[1])
            form_s[key] = wall(lambda: o(B_local, feature_chunks=key[1]), other_steps)
        # the two parts of every form alone, each on the compute stream (nothing overlaps here): what a
        # step of the form costs when exchange and kernels run back to back
        for (e, c), o in forms.items():
            o.exchange_only(B_local, c)
            exch_ms[(e, c)] = event_ms(lambda: o.exchange_only(B_local, c), other_steps)
            operand = o.exchange_only(B_local, c)
            o.spmm_only(operand)
            kern_by_exchange[(e, c)] = event_ms(lambda: o.spmm_only(operand), other_steps)

    kern_ms_own = kern_ms  # this rank's own kernel time (kern_ms becomes the slowest rank's below)
    vec = [elapsed, kern_ms] + [form_s[k] for k in forms] + [exch_ms[e] for e in exch_ms] + [kern_by_exchange[e] for e in kern_by_exchange]
    t = torch.tensor(vec, dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    vec = [float(x) for x in t]
    elapsed, kern_ms = vec[0], vec[1]
    form_s = dict(zip(forms, vec[2:2 + len(forms)]))
    exch_ms = dict(zip(exch_ms, vec[2 + len(forms):2 + len(forms) + len(exch_ms)]))
    kern_by_exchange = dict(zip(kern_by_exchange, vec[2 + len(forms) + len(exch_ms):]))
    recv_rows = {}
    shard_sizes = None
    esz = B_local.element_size()
    minmax = args.op in ("spmm_max", "spmm_min")
    if half:  # 2-byte dense rows: 2 F of every (12 + 2 F) bytes per edge, 8 + 2 F per row
        my_alg = nnz * (8 + 4 + 2 * F) + M * (8 + 2 * F) + (M * F * 8 if minmax else 0)
    else:
        my_alg = algorithmic_bytes(nnz, M, F, True, minmax)
    roof_rank, roof_ms, roof_alg, roof_nnz = 0, kern_ms_own, my_alg, nnz
    if use_dist:
        # the roofline pairs ONE rank's kernel time with that same rank's bytes: the slowest rank's
        kr = torch.zeros(world, 3, dtype=torch.float64, device=device)
        kr[rank, 0], kr[rank, 1], kr[rank, 2] = kern_ms_own, float(my_alg), float(nnz)
        dist.all_reduce(kr)
        roof_rank = int(kr[:, 0].argmax())
        roof_ms, roof_alg, roof_nnz = float(kr[roof_rank, 0]), int(kr[roof_rank, 1]), int(kr[roof_rank, 2])
        rr = torch.tensor([o.rows_received_per_step() for o in ops_by_exchange.values()], dtype=torch.int64, device=device)
        dist.all_reduce(rr, op=dist.ReduceOp.MAX)
        recv_rows = dict(zip(ops_by_exchange, rr.tolist()))
        sz = torch.zeros(world, 2, dtype=torch.int64, device=device)
        sz[rank, 0], sz[rank, 1] = M, nnz
        dist.all_reduce(sz)
        shard_sizes = sz.tolist()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_nnz / (elapsed / args.steps) / 1e9
        alg = roof_alg
        achieved = alg / (roof_ms * 1e-3) / 1e9
        traffic, traffic_note = None, "not collected for this configuration"
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists() and world == 1 and args.config == "c3" and not half:
            rec = json.loads(tfile.read_text()).get(f"{args.op}_c3", {})
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_note = (f"read from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round "
                            f"{rec.get('round', '?')}, {rec.get('source', '?')}); not measured in this run")
        if use_dist and scaling == "strong":
            what = (f"{args.op} fwd, the ONE uniform random CSR {M_PER_GPU}x{N} with nnz={total_nnz} "
                    f"({'a tenth of BASELINE config 3: rehearsal' if rehearse else 'BASELINE config 3'}), "
                    f"dense F={F} {args.dtype}, rows split by nnz over {world} GPU(s)")
        else:
            if world == 1 and args.config == "c3":  # the base of the strong series: the ONE config-3 matrix on one GPU
                what = (f"{args.op} fwd, uniform random CSR {M}x{N}, nnz={nnz}, dense F={F} {args.dtype} "
                        f"(BASELINE config 3, whole on one GPU)")
            else:
                cfg_name = "BASELINE config 3 per GPU" if args.config == "c3" else "BASELINE config 4's per-GPU share (F = 256)"
                what = (f"{args.op} fwd, uniform random CSR {M}x{N} per GPU, nnz={nnz} per GPU, "
                        f"dense F={F} {args.dtype} ({cfg_name})")
        if use_dist:
            what += (f"; + exchange of B ({N}x{F}, row-sharded) every step: {chosen[0]}"
                     + (f", in {chosen[1]} column slices overlapped with the SpMM" if chosen[1] > 1 else ""))
        line = {
            "metric": f"{args.op}_gedges_per_s",
            "value": round(value, 4),
            "unit": "GEdges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": scaling,  # N = 1 is the base of the series the flags select (strong for config 3: total work fixed as N grows)
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": what,
                "rows_per_gpu": M, "nnz_per_gpu": nnz, "feat": F, "index_dtype": "int64",
                "variant": args.variant,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "spmm_half_row_kernel" if half else "spmm_fused_kernel",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": alg,
                "kernel_ms": round(roof_ms, 4),
                "kernel_gedges_per_s": round(roof_nnz / (roof_ms * 1e-3) / 1e9, 4),
            },
        }
        if rehearse:
            line["rehearsal"] = (f"{world} ranks sharing ONE GPU over gloo (PSA_BENCH_REHEARSE_ON_ONE_GPU=1): a functional run of "
                                 "the N > 1 path; none of its times is a measurement")
        if use_dist:
            rate = lambda s_: round(total_nnz / s_ / 1e9, 4)  # noqa: E731
            serial = {e: rate(form_s[(e, 1)]) for e in ops_by_exchange if (e, 1) in form_s}
            overl = {e: rate(form_s[(e, c)]) for (e, c) in form_s if c > 1}
            line["config"]["rows_nnz_by_rank"] = shard_sizes
            line["roofline"]["note"] = (f"rank {roof_rank}'s block (the slowest rank's kernel): its own algorithmic bytes "
                                        "over its own HIP-event kernel time")
            line["multi_gpu"] = {
                "form_timed_as_value": f"{chosen[0]}/chunks{chosen[1]}",
                "form_chosen_by": "flags" if len(forms) == 1 else f"{TUNE_STEPS} steady-state steps per form before the warmup (ms per step by form)",
                "tune_ms_per_step_by_form": tune_ms or None,
                "step_ms_by_form": {f"{e}/chunks{c}": round(x * 1e3, 4) for (e, c), x in form_s.items()},
                "local_plan": "raw arrays, algo auto" if args.no_plan else op.local_storage()._spmm_algo(),
                "spmm_only_aggregate_gedges_per_s": rate(kern_ms * 1e-3),
                "end_to_end_serial_gedges_per_s": serial,
                "end_to_end_overlapped_gedges_per_s": overl,
                "overlapped_feature_chunks": 4,
                "exchange_ms": {e: round(x, 4) for (e, c), x in exch_ms.items() if c == 1},
                "exchange_ms_by_form": {f"{e}/chunks{c}": round(x, 4) for (e, c), x in exch_ms.items()},
                "kernel_ms_by_form": {f"{e}/chunks{c}": round(x, 4) for (e, c), x in kern_by_exchange.items()},
                "kernel_plus_exchange_ms_by_form": {f"{e}/chunks{c}": round(exch_ms[(e, c)] + kern_by_exchange[(e, c)], 4)
                                                    for (e, c) in exch_ms},
                "step_over_parts_by_form": {f"{e}/chunks{c}": round(form_s[(e, c)] * 1e3 / (exch_ms[(e, c)] + kern_by_exchange[(e, c)]), 4)
                                            for (e, c) in exch_ms},
                "allgather_bytes_received_per_rank": recv_rows.get("full", 0) * F * esz,
                "bytes_received_per_rank_by_exchange": {e: r * F * esz for e, r in recv_rows.items()},
                "buffer_bytes_held_by_form_objects": {e: o.buffer_bytes() for e, o in ops_by_exchange.items()},
                "note": "value counts the exchange of B inside every step, in the form named by form_timed_as_value; "
                        "spmm_only_* is the local-kernel rate with B already assembled (slowest rank)",
            }
        if (not args.no_cpu and args.op == "spmm_sum" and world == 1 and isinstance(B_full, torch.Tensor) and not half
                and not (use_dist and chosen[0] == "halo")):  # CPU leg: N = 1 only
            info, ref, rows = cpu_baseline(rowptr, col, val, B_full)
            got = out[:rows].cpu().numpy()
            scale = np.abs(ref).max()
            line["cpu_baseline"] = info
            line["check_max_abs_err_vs_oracle"] = float(np.abs(got - ref).max() / scale)
        if "cpu_baseline" not in line:
            line["cpu_baseline"] = None
            line["cpu_baseline_note"] = ("the CPU leg runs on rank 0 at N = 1 with --op spmm_sum --dtype f32 only (here: "
                                         f"n_gpus={world}, op={args.op}, dtype={args.dtype}, no_cpu={args.no_cpu})")
        if not args.no_extra and world == 1 and not use_dist and args.config == "c3" and not half:
            # after the timed region: the rest of BASELINE config 3 (mean / max forward, sum / max
            # forward + backward) and the forward on a power-law graph; headline fields unchanged
            line["c3_other_ops"] = breadth(rowptr, col, val, B_full, max(5, args.steps // 5))
            line["c2_spmm_sum_fwd"] = config2(device)
            del out
            line["power_law"] = power_law(device, F, max(5, args.steps // 5))
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
else:  # imported (tools/, tests/perf/): the helpers above need numpy / torch; only the launcher process goes without
    load_compute_modules()
