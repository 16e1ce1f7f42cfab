#!/usr/bin/env python3
"""Headline benchmark: SpMM GEdges/s + achieved-HBM fraction.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--op spmm_sum]

N = 1 (default): BASELINE.json config 3 — random CSR, 2M x 2M, nnz = 20M,
dense F = 128 fp32, one `spmm_sum` forward per step, inputs resident in HBM.
N > 1 (launched by torch.distributed.run, one rank per GPU): the same per-GPU
rows/edges on every rank (weak scaling, BASELINE config 4's structure): rank r
owns rows [r*2M, (r+1)*2M) of A (20M edges, columns over all N*2M nodes) and
the matching 2M-row block of B; a step = RCCL all-gather of B + local SpMM
(paddle_sparse_amd.distributed.RowPartitionedSpMM).

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

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

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


def cpu_baseline(rowptr, col, val, B, budget_s: float = 12.0):
    """Oracle C port (OpenMP over rows) timed on this box's host cores on a
    bounded row-prefix of the same workload; also used to check the GPU."""
    import oracle

    threads = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    M = rowptr.numel() - 1
    sample_rows = min(M, 1_000_000)
    rp = rowptr[: sample_rows + 1].cpu().numpy()
    e = int(rp[-1])
    c = col[:e].cpu().numpy()
    v = val[:e].cpu().numpy()
    Bh = B.cpu().numpy()
    oracle.spmm("sum", rp[:1001], c, v, Bh, threads=threads)  # warm the pool
    best, reps, t_all = float("inf"), 0, time.perf_counter()
    out = None
    while reps < 5 and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        out, _ = oracle.spmm("sum", rp, c, v, Bh, threads=threads)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    info = {
        "value": round(e / best / 1e9, 4),
        "unit": "GEdges/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {sample_rows} rows ({e} edges) of the same CSR x the full B, "
                  f"oracle_spmm_omp, best of {reps}",
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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--op", default="spmm_sum", choices=["spmm_sum", "spmm_mean", "spmm_max", "spmm_min"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--feature-chunks", type=int, default=0,
                    help="N > 1: all-gather B in this many column slices and run the SpMM of a "
                         "slice under the exchange of the next ones (1: one all-gather, then the SpMM; "
                         "default 0: time both forms during warmup and keep the faster one)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
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
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    M, nnz, F = M_PER_GPU, NNZ_PER_GPU, FEAT
    N = M * world
    reduce = args.op.split("_", 1)[1]
    rowptr, col, val = make_workload(M, N, nnz, F, seed=2 + rank, device=device)
    g = torch.Generator(device=device).manual_seed(100 + rank)
    B_local = torch.randn(M, F, generator=g, device=device)
    ops.spmm_set_variant(args.variant)

    chunks = 1
    if use_dist:
        from paddle_sparse_amd.distributed import RowPartitionedSpMM, RowShard

        op = RowPartitionedSpMM(RowShard(rowptr, col, val, rank * M, (rank + 1) * M, N), reduce=reduce)
        chunks = args.feature_chunks
        step = lambda: op(B_local, feature_chunks=chunks)  # noqa: E731  all-gather(B) + local HIP SpMM
        B_full = op.gather(B_local)
    else:
        fn = getattr(ops, args.op)
        B_full = B_local
        step = lambda: fn(rowptr, col, val, B_full)  # noqa: E731

    def local_kernel():
        return ops._spmm(reduce, rowptr, col, val, B_full)[0]

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Two forms of the same step exist (one all-gather then the SpMM; or column
    # slices whose exchanges overlap the SpMM of earlier slices).  Which is faster
    # depends on the fabric, so — untimed, before the warmup — both run a few
    # steps, the slowest rank's times decide, and every rank keeps the same form.
    overlap_chunks = 4
    tune_ms = {}
    if use_dist and chunks == 0:
        candidates = [1] + ([overlap_chunks] if F % overlap_chunks == 0 else [])
        spent = []
        for c in candidates:
            op(B_local, feature_chunks=c)
            sync_all()
            t_c = time.perf_counter()
            for _ in range(3):
                op(B_local, feature_chunks=c)
            sync_all()
            spent.append((time.perf_counter() - t_c) / 3)
        spent_t = torch.tensor(spent, dtype=torch.float64, device=device)
        dist.all_reduce(spent_t, op=dist.ReduceOp.MAX)
        spent = [float(x) for x in spent_t]
        chunks = candidates[spent.index(min(spent))]
        tune_ms = {str(c): round(x * 1e3, 4) for c, x in zip(candidates, spent)}

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
    gather_ms = event_ms(lambda: op.gather(B_local), max(3, args.steps // 5)) if use_dist else 0.0

    # SURVEY.md §8(e) asks for both end-to-end figures: the form that was NOT
    # timed above runs a few steps here, so the line carries serial and overlapped.
    other_chunks = overlap_chunks if chunks <= 1 else 1
    other_s, other_steps = 0.0, 0
    if use_dist and F % overlap_chunks == 0:
        other_steps = max(3, args.steps // 5)
        op(B_local, feature_chunks=other_chunks)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(other_steps):
            op(B_local, feature_chunks=other_chunks)
        sync_all()
        other_s = time.perf_counter() - t1

    t = torch.tensor([elapsed, kern_ms, gather_ms, other_s], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms, gather_ms, other_s = (float(x) for x in t)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * nnz / (elapsed / args.steps) / 1e9
        alg = algorithmic_bytes(nnz, M, F, True, args.op in ("spmm_max", "spmm_min"))
        achieved = alg / (kern_ms * 1e-3) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists() and world == 1:
            traffic = json.loads(tfile.read_text()).get(f"{args.op}_c3", {}).get("hbm_bytes_per_launch")
        line = {
            "metric": f"{args.op}_gedges_per_s",
            "value": round(value, 4),
            "unit": "GEdges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.op} fwd, uniform random CSR {M}x{N} per GPU, nnz={nnz} per GPU, "
                            f"dense F={F} fp32 (BASELINE config 3 per GPU)"
                            + (f"; + RCCL all-gather of B ({N}x{F}) every step"
                               + (f", in {chunks} column slices overlapped with the SpMM" if chunks > 1 else "")
                               if use_dist else ""),
                "rows_per_gpu": M, "nnz_per_gpu": nnz, "feat": F, "index_dtype": "int64",
                "variant": args.variant,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "spmm_fused_kernel",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg,
                "kernel_ms": round(kern_ms, 4),
                "kernel_gedges_per_s": round(nnz / (kern_ms * 1e-3) / 1e9, 4),
            },
        }
        if use_dist:
            other = round(world * nnz / (other_s / other_steps) / 1e9, 4) if other_steps else None
            serial, overlapped = (value, other) if chunks <= 1 else (other, value)
            line["multi_gpu"] = {
                "allgather_ms": round(gather_ms, 4),
                "feature_chunks": chunks,
                "feature_chunks_chosen_by": "flag" if args.feature_chunks else "warmup timing (ms per step by form)",
                "warmup_ms_per_step_by_feature_chunks": tune_ms or None,
                "allgather_bytes_received_per_rank": (world - 1) * M * F * 4,
                "spmm_only_aggregate_gedges_per_s": round(world * nnz / (kern_ms * 1e-3) / 1e9, 4),
                "end_to_end_serial_gedges_per_s": None if serial is None else round(serial, 4),
                "end_to_end_overlapped_gedges_per_s": None if overlapped is None else round(overlapped, 4),
                "overlapped_feature_chunks": overlap_chunks,
                "note": "value counts the exchange of B inside every step, in the form named by feature_chunks "
                        "(1 = one all-gather then the SpMM); spmm_only_* is the local-kernel rate with B "
                        "already assembled",
            }
        if not args.no_cpu and args.op == "spmm_sum" and world == 1:  # CPU leg: N = 1 only
            info, ref, rows = cpu_baseline(rowptr, col, val, B_full)
            got = out[:rows].cpu().numpy()
            scale = np.abs(ref).max()
            line["cpu_baseline"] = info
            line["check_max_abs_err_vs_oracle"] = float(np.abs(got - ref).max() / scale)
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
