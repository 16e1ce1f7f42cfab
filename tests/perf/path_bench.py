#!/usr/bin/env python3
"""One line per SURVEY.md §8(a) row: GPU time on one MI355X, the algorithmic
bytes of §8(d) and the HBM fraction they imply, and the CPU baseline of
§8(d) timed on this box's host cores (oracle/ C restatements: single thread —
the reference's CPU ops carry no OpenMP pragma — and all host threads).

Workloads: C3 (M = N = 2M, nnz = 20M uniform, F = 128) for every row, plus the
measured HBM yardsticks (device copy and triad) the fractions can be read
against.  `--cpu-edges` bounds the CPU sample for the sort rows.

    python tests/perf/path_bench.py > gpurun_out/path_bench.txt
"""
import argparse
import os
import sys
import time
from pathlib import Path



def host_threads() -> int:
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


# before any OpenMP runtime starts: the box shows 256 CPUs but grants 16
os.environ["OMP_NUM_THREADS"] = str(host_threads())

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402  (CPU baseline leg only)
import paddle_sparse_amd as ps  # noqa: E402
from paddle_sparse_amd import SparseStorage, SparseTensor, ops  # noqa: E402
from paddle_sparse_amd.reduce import reduction  # noqa: E402

PEAK = 8.0e12


def gpu_ms(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def cpu_ms(fn, reps=1):
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=2_000_000)
ap.add_argument("--edges", type=int, default=20_000_000)
ap.add_argument("--features", type=int, default=128)
ap.add_argument("--cpu-edges", type=int, default=20_000_000)
ap.add_argument("--no-cpu", action="store_true")
args = ap.parse_args()
M = N = args.nodes
nnz, F = args.edges, args.features
T = host_threads()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(2)
row_u = rng.integers(0, M, nnz)
col_u = rng.integers(0, N, nnz)
val_h = rng.standard_normal(nnz, dtype=np.float32)
B_h = rng.standard_normal((N, F), dtype=np.float32)
g_h = rng.standard_normal((M, F), dtype=np.float32)

row_d, col_d = torch.from_numpy(row_u).to(dev), torch.from_numpy(col_u).to(dev)
val_d, B_d, g_d = torch.from_numpy(val_h).to(dev), torch.from_numpy(B_h).to(dev), torch.from_numpy(g_h).to(dev)
index_d = torch.stack([row_d, col_d])

lines = []


def report(rowid, what, ms, alg_bytes, cpu1=None, cpuT=None, note=""):
    gbs = alg_bytes / ms / 1e6
    s = f"{rowid:4s} {what:44s} {ms:9.3f} ms {alg_bytes / 1e9:8.3f} GB {gbs:8.0f} GB/s {100 * gbs * 1e9 / PEAK:5.1f}%"
    if cpu1 is not None:
        s += f" | cpu 1t {cpu1:9.1f} ms"
    if cpuT is not None:
        s += f" {T}t {cpuT:9.1f} ms ({cpuT / ms:6.0f}x)"
    if note:
        s += "  " + note
    lines.append(s)
    print(s, flush=True)


print(f"# path_bench: M=N={M} nnz={nnz} F={F}; host threads {T} (os.cpu_count {os.cpu_count()}); "
      f"{torch.cuda.get_device_name(0)}")
print("# row  op                                            gpu time   alg bytes   rate   of 8 TB/s | CPU baseline (oracle/, same inputs)")

# ---- yardsticks: what this card streams ------------------------------------
a = torch.empty(256 << 20, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
c = torch.empty_like(a)
report("-", "device copy 1 GiB -> 1 GiB (torch copy_)", gpu_ms(lambda: b.copy_(a)), 2 * a.numel() * 4)
report("-", "triad c = a + s*b, 3 x 1 GiB (torch add)", gpu_ms(lambda: torch.add(a, b, alpha=2.0, out=c)), 3 * a.numel() * 4)
del a, b, c
torch.cuda.empty_cache()

# ---- a1 / a3 / a4: ctor sort, coalesce --------------------------------------
n_cpu = min(nnz, args.cpu_edges)
passes = ((M * N - 1).bit_length() + 7) // 8
sort_model = nnz * 24 + nnz * (32 * passes - 4)      # keys + histogram read + P x (key+idx32 in/out)


def cpu_coalesce(threads):
    return cpu_ms(lambda: oracle.coalesce_c(row_u[:n_cpu], col_u[:n_cpu], val_h[:n_cpu], M, N, "add", threads))


c1 = cT = None
if not args.no_cpu:
    c1, cT = cpu_coalesce(1) * nnz / n_cpu, cpu_coalesce(T) * nnz / n_cpu
ms = gpu_ms(lambda: SparseStorage(row=row_d, col=col_d, value=val_d, sparse_sizes=(M, N), is_sorted=False), reps=5)
report("a1", "SparseStorage(row, col, value) unsorted", ms, nnz * 20 * 2)
ms = gpu_ms(lambda: ps.coalesce(index_d, val_d, M, N, "add"), reps=5)
idx_c, val_c = ps.coalesce(index_d, val_d, M, N, "add")
nnz2 = idx_c.shape[1]
report("a4", f"coalesce(index, value, m, n) -> {nnz2} (floor model)", ms, nnz * 20 + nnz2 * 20, c1, cT)
report("a4", "  same, LSD-radix traffic model", ms, sort_model + nnz * 28 + nnz2 * 28)
st_dup = SparseStorage(row=row_d, col=col_d, value=val_d, sparse_sizes=(M, N), is_sorted=False)
ms = gpu_ms(lambda: st_dup.coalesce("add"), reps=5)
report("a3", "SparseStorage.coalesce() on sorted duplicates", ms, nnz * 20 + nnz2 * 20)

# ---- a2: index_sort -----------------------------------------------------------
keys_d, _ = ops.make_keys(row_d, col_d, N)
keys_h = row_u * N + col_u
if not args.no_cpu:
    c1 = cpu_ms(lambda: oracle.index_sort_c(keys_h[:n_cpu], M * N, 1)) * nnz / n_cpu
    cT = cpu_ms(lambda: oracle.index_sort_c(keys_h[:n_cpu], M * N, T)) * nnz / n_cpu
ms = gpu_ms(lambda: ops.index_sort(keys_d, M * N, with_sorted_inputs=True), reps=5)
report("a2", f"index_sort {passes} x 8-bit passes (floor: in + out)", ms, nnz * 8 + nnz * 16, c1, cT)
report("a2", "  same, LSD-radix traffic model", ms, sort_model - nnz * 16)
del keys_d, st_dup

# ---- coalesced operand for the rest ---------------------------------------------
row_s, col_s = idx_c[0].contiguous(), idx_c[1].contiguous()
row_sh, col_sh, val_sh = row_s.cpu().numpy(), col_s.cpu().numpy(), val_c.cpu().numpy()
E = nnz2

# ---- a5 / a6 / a7 ----------------------------------------------------------------
c1 = None if args.no_cpu else cpu_ms(lambda: oracle.ind2ptr(row_sh, M), 3)
ms = gpu_ms(lambda: ops.ind2ptr(row_s, M))
report("a5", "ind2ptr(row, M)", ms, E * 8 + (M + 1) * 8, c1)
rowptr_d = ops.ind2ptr(row_s, M)
rowptr_h = rowptr_d.cpu().numpy()
c1 = None if args.no_cpu else cpu_ms(lambda: oracle.ptr2ind(rowptr_h, E), 3)
ms = gpu_ms(lambda: ops.ptr2ind(rowptr_d, E))
report("a6", "ptr2ind(rowptr, E)", ms, (M + 1) * 8 + E * 8, c1)


def fresh():
    return SparseStorage(row=row_s, rowptr=rowptr_d, col=col_s, value=val_c, sparse_sizes=(M, N),
                         is_sorted=True, trust_data=True)


ms = gpu_ms(lambda: fresh().rowcount())
report("a7", "rowcount() from rowptr", ms, (M + 1) * 8 + M * 8)

# ---- a8 / a9 -----------------------------------------------------------------------
ms = gpu_ms(lambda: fresh().colcount())
report("a8", "colcount() cold (3-pass stable sort of col + ind2ptr)", ms, E * 8 + N * 8)
ms = gpu_ms(lambda: fresh().colptr())
report("a8", "colptr() cold (same; leaves csr2csc behind)", ms, E * 8 + (N + 1) * 8)
key_csc = col_sh * M + row_sh
if not args.no_cpu:
    c1 = cpu_ms(lambda: oracle.index_sort_c(key_csc, M * N, 1))
    cT = cpu_ms(lambda: oracle.index_sort_c(key_csc, M * N, T))
ms = gpu_ms(lambda: fresh().csr2csc(), reps=5)
report("a9", "csr2csc() = stable index_sort of col alone", ms, E * 8 + E * 8, c1, cT)
st = fresh()
st.csr2csc()


def csc2csr():
    st._csc2csr = None
    return st.csc2csr()


ms = gpu_ms(csc2csr)
report("a9", "csc2csr() = inverse permutation (scatter)", ms, E * 16)

# ---- a10 / a12 -----------------------------------------------------------------------
A = SparseTensor.from_storage(fresh())
ms = gpu_ms(lambda: SparseTensor.from_storage(fresh()).t(), reps=5)
report("a10", "SparseTensor.t() cold (col sort, row + value gathers)", ms, E * 16 + E * 24 + E * 16)
A.storage.csr2csc()
A.storage.colptr()


def drop_value_memo():
    # value[csr2csc] is memoised on the storage for as long as the value tensor is unchanged
    # (storage._value_in_csc_order): a call that finds it does no work and is not a measurement
    # of the gather.  The memo is dropped inside the timed call; the structure caches stay.
    A.storage._value_csc_memo = None


def t_warm():
    drop_value_memo()
    return A.t()


ms = gpu_ms(t_warm)
report("a10", "SparseTensor.t() structure cached (value gather)", ms, E * 16)
if not args.no_cpu:
    c1 = cpu_ms(lambda: oracle.coalesce_c(col_u[:n_cpu], row_u[:n_cpu], val_h[:n_cpu], N, M, "add", 1)) * nnz / n_cpu
    cT = cpu_ms(lambda: oracle.coalesce_c(col_u[:n_cpu], row_u[:n_cpu], val_h[:n_cpu], N, M, "add", T)) * nnz / n_cpu
ms = gpu_ms(lambda: ps.transpose(index_d, val_d, M, N), reps=5)
report("a10", "transpose(index, value, m, n) (floor model)", ms, nnz * 20 + nnz2 * 20, c1, cT)


def csc_warm():
    drop_value_memo()
    return A.csc()


ms = gpu_ms(csc_warm)
report("a12", "csc() structure cached (value gather)", ms, E * 16)

# ---- a11 --------------------------------------------------------------------------------
for reduce in ("sum", "max"):
    ms = gpu_ms(lambda: reduction(A, 1, reduce))
    report("a11", f"reduction(dim=1, {reduce}) segment over rowptr", ms, E * 4 + (M + 1) * 8 + M * 4)
    ms = gpu_ms(lambda: reduction(A, 0, reduce))
    report("a11", f"reduction(dim=0, {reduce}) CSC cached: segment path", ms, E * 12 + (N + 1) * 8 + N * 4)
    # cold: a storage without caches built inside every timed call (a tensor kept across calls has them after the first)
    ms = gpu_ms(lambda: reduction(SparseTensor.from_storage(fresh()), 0, reduce), reps=5)
    report("a11", f"reduction(dim=0, {reduce}) cold: col sort + value gather + segment path", ms, E * 8 + E * 16 + E * 12 + N * 4)
ms = gpu_ms(lambda: reduction(A, None, "sum"))
report("a11", "reduction(dim=None, sum)", ms, E * 4)

# ---- a13: SpMM forward / backward ----------------------------------------------------------
fw_bytes = E * (8 + 4 + 4 * F) + M * (8 + 4 * F)
if not args.no_cpu:
    # bounded sample: the first rows holding ~1/8 of the edges, scaled
    r_cut = int(np.searchsorted(rowptr_h, E // 8))
    e_cut = int(rowptr_h[r_cut])
    c1 = cpu_ms(lambda: oracle.spmm("sum", rowptr_h[:r_cut + 1], col_sh[:e_cut], val_sh[:e_cut], B_h, 1)) * E / e_cut
    oracle.spmm("sum", rowptr_h[:1001], col_sh[:e_cut], val_sh[:e_cut], B_h, T)  # start the thread pool
    cT = cpu_ms(lambda: oracle.spmm("sum", rowptr_h[:r_cut + 1], col_sh[:e_cut], val_sh[:e_cut], B_h, T), 3) * E / e_cut
for reduce in ("sum", "mean", "min", "max"):
    fn = getattr(ops, f"spmm_{reduce}")
    ms = gpu_ms(lambda: fn(rowptr_d, col_s, val_c, B_d), reps=20)
    if reduce in ("sum", "mean"):
        report("a13", f"spmm_{reduce} forward", ms, fw_bytes, c1 if reduce == "sum" else None,
               cT if reduce == "sum" else None)
    else:
        report("a13", f"spmm_{reduce} forward (+ arg_out)", ms, fw_bytes + M * F * 8)
ms = gpu_ms(lambda: ops.spmm_value_bw(row_s, rowptr_d, col_s, B_d, g_d, "sum"), reps=20)
# bytes the pass moves: col + the gathered mat row + grad_value per entry, rowptr + the row's own grad row per row.
# (SURVEY.md 8(d)'s SDDMM model — 8 + 8 + 8 F + 4 per entry, 20.9 GB — reads the grad row per ENTRY; this kernel holds
# it in registers per row, so against that model it would read as 122 % of the HBM peak.)
report("a13", "spmm_value_bw (grad of value, sum; survey model 20.9 GB)", ms, E * (8 + 4 * F + 4) + M * (8 + 4 * F))
csr2csc, colptr_d = A.storage.csr2csc(), A.storage.colptr()
row_csc = A.storage._row_in_csc_order()
ms = gpu_ms(lambda: ops.transpose_weights(val_c, csr2csc, None, None, False), reps=20)
report("a13", "grad of mat: value[csr2csc] gather", ms, E * 16)
w_t = ops.transpose_weights(val_c, csr2csc, None, None, False)
ms = gpu_ms(lambda: ops.spmm_sum(colptr_d, row_csc, w_t, g_d), reps=20)
report("a13", "grad of mat: SpMM over CSC", ms, E * (8 + 4 + 4 * F) + N * (8 + 4 * F))
out, arg = ops.spmm_max(rowptr_d, col_s, val_c, B_d)
ms = gpu_ms(lambda: ops.spmm_minmax_bw(col_s, val_c, B_d, g_d, arg, True, True), reps=10)
report("a13", "spmm_max backward, float atomics (both grads)", ms, M * F * (4 + 8 + 4 + 4 + 4 + 4) + N * F * 4 + E * 4)
tags = A.storage._csc_edge_tags()
ms = gpu_ms(lambda: ops.spmm_minmax_bw_csc(rowptr_d, colptr_d, row_csc, csr2csc, tags, val_c, B_d, g_d, arg), reps=10)
report("a13", "spmm_max backward, one CSC pass (both grads)", ms,
       M * F * 9 + E * (4 * F + F + 21) + N * F * 8 + E * 4)
# what autograd runs: the forward leaves the row-local arg_out (1 byte per element here), no int64 arg_out, no compress pass
_, _, local = ops._spmm("max", rowptr_d, col_s, val_c, B_d, want_arg_bytes=True, want_arg=False)
inv = A.storage.csc2csr()
ms = gpu_ms(lambda: ops.spmm_minmax_bw_csc(rowptr_d, colptr_d, row_csc, csr2csc, tags, val_c, B_d, g_d, None, csc2csr=inv,
                                           arg_bytes=local), reps=10)
report("a13", "spmm_max backward, one CSC pass fed by the forward's row-local arg (both grads)", ms,
       E * (8 + 8 + 4 + 4 * F + F + 1 + 4 + 16) + N * (8 + 8 * F))
ms = gpu_ms(lambda: ops.spmm_minmax_bw_csc(rowptr_d, colptr_d, row_csc, csr2csc, tags, val_c, B_d, g_d, None, want_value=False,
                                           arg_bytes=local), reps=10)
report("a13", "  same, grad of mat only (fixed adjacency)", ms, E * (8 + 8 + 4 + 4 * F + F + 1) + N * (8 + 4 * F))
ms = gpu_ms(lambda: ops.spmm_sum_bw_csc(colptr_d, row_csc, csr2csc, val_c, B_d, g_d, True, csc2csr=inv), reps=10)
report("a13", "spmm_sum backward, one CSC pass (both grads), gathers through csr2csc / csc2csr", ms, E * (8 + 8 + 4 + 4 * F + 4 + 16) + N * (8 + 8 * F))
# what autograd runs from the second step on: value[csr2csc] and grad_value's way back along planned routes
to_csc, to_csr = A.storage._permute_plan("to_csc", force=True), A.storage._permute_plan("to_csr", force=True)


def sum_bw_planned():
    return ops.spmm_sum_bw_csc(colptr_d, row_csc, csr2csc, val_c, B_d, g_d, True, csc2csr=inv, to_csr_plan=to_csr,
                               value_csc=ops.permute_apply(val_c, to_csc))


def max_bw_planned():
    return ops.spmm_minmax_bw_csc(rowptr_d, colptr_d, row_csc, csr2csc, tags, val_c, B_d, g_d, None, csc2csr=inv,
                                  arg_bytes=local, to_csr_plan=to_csr, value_csc=ops.permute_apply(val_c, to_csc))


ms = gpu_ms(sum_bw_planned, reps=10)
report("a13", "spmm_sum backward as autograd runs it: planned value[csr2csc] + pass + planned way back", ms,
       E * (8 + 8 + 4 + 4 * F + 4 + 16) + N * (8 + 8 * F))
ms = gpu_ms(max_bw_planned, reps=10)
report("a13", "spmm_max backward as autograd runs it (row-local arg, planned routes)", ms,
       E * (8 + 8 + 4 + 4 * F + F + 1 + 4 + 16) + N * (8 + 8 * F))
ms = gpu_ms(lambda: ops.permute_apply(val_c, to_csc), reps=20)
report("a13", "  value[csr2csc] along the planned route (two streaming passes)", ms, E * 24)
