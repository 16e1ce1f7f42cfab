"""Row-partitioned multi-GPU SpMM: one process per GPU, RCCL over xGMI.

Not in the reference (it has no distributed code at all);
this is the north-star's multi-GPU path:

  * the sparse A is split into `world` contiguous row blocks, balanced by nnz
    through rowptr (each rank keeps its rebased rowptr slice and its col/value
    slice, and is the only writer of out[r0:r1, :] — no output exchange);
  * the dense B is row-sharded in equal blocks (rank r owns rows
    [r*nb, (r+1)*nb), nb = ceil(N / world)); every step reassembles it with
    ONE all-gather (`torch.distributed.all_gather_into_tensor`, backend "nccl"
    = RCCL on ROCm) and runs the local HIP SpMM on the full B;
  * coalesce / index_sort / ind2ptr stay single-GPU (replicas only).

xGMI arithmetic that decides what this can reach (8-GPU full mesh, 7 links x
~153 GB/s per GPU): every rank must receive (world-1)/world of B every step;
at N = 16M, F = 128 that is 7.2 GB per rank, >= 6.7 ms even at the full
per-GPU ingest rate, against ~2 ms of local SpMM — the exchange, not the
kernel, bounds the step.  bench.py reports both.

Everything here except the local kernel call is index arithmetic and one
collective, so it runs on CPU tensors with the gloo backend too (that is how
tests/test_distributed.py covers it at world_size 2).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def partition_rows_by_nnz(rowptr: torch.Tensor, world: int) -> List[int]:
    """Row boundaries b[0..world] (b[0] = 0, b[world] = M) such that every
    block [b[r], b[r+1]) holds about nnz / world edges."""
    M = rowptr.numel() - 1
    nnz = int(rowptr[-1])
    targets = torch.tensor([(nnz * r) // world for r in range(1, world)],
                           dtype=rowptr.dtype, device=rowptr.device)
    cuts = []
    if world > 1:
        rp = rowptr.contiguous()
        hi = torch.searchsorted(rp, targets, right=False).clamp_(0, M)  # first rowptr >= target
        lo = (hi - 1).clamp_(min=0)
        # a row boundary on either side of the target: take the nearer one
        take_lo = (targets - rp[lo]) < (rp[hi] - targets)
        cuts = torch.where(take_lo, lo, hi).tolist()
    bounds = [0] + [min(max(int(c), 0), M) for c in cuts] + [M]
    for i in range(1, len(bounds)):  # keep it monotone when rows are huge
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def partition_rows_evenly(M: int, world: int) -> List[int]:
    return [(M * r) // world for r in range(world)] + [M]


def dense_block_rows(N: int, world: int) -> int:
    """Rows of B each rank owns (the last rank's block may be partly padding)."""
    return (N + world - 1) // world


@dataclass
class RowShard:
    """Rank-local piece of a row-partitioned CSR matrix."""
    rowptr: torch.Tensor          # int64[m_local + 1], rebased to start at 0
    col: torch.Tensor             # int64[nnz_local], GLOBAL column ids
    value: Optional[torch.Tensor]  # f32[nnz_local] or None
    row_begin: int
    row_end: int
    num_cols: int                 # N of the global matrix

    @property
    def num_rows(self) -> int:
        return self.row_end - self.row_begin

    @property
    def nnz(self) -> int:
        return self.col.numel()


def shard_csr(rowptr: torch.Tensor, col: torch.Tensor, value: Optional[torch.Tensor],
              num_cols: int, bounds: List[int], rank: int) -> RowShard:
    """Cut rank `rank`'s row block out of a CSR that this process holds whole."""
    r0, r1 = bounds[rank], bounds[rank + 1]
    e0, e1 = int(rowptr[r0]), int(rowptr[r1])
    local_ptr = (rowptr[r0:r1 + 1] - e0).contiguous()
    return RowShard(local_ptr, col[e0:e1].contiguous(),
                    None if value is None else value[e0:e1].contiguous(), r0, r1, num_cols)


def all_gather_dense(b_local: torch.Tensor, num_rows: int, group=None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Reassemble B [num_rows, F] from equal row blocks (one per rank).

    b_local is this rank's block: [dense_block_rows(num_rows, world), F]
    (rows past num_rows on the last rank are padding and are dropped).
    `out` may hold a reusable [world * block, F] buffer."""
    world = dist.get_world_size(group)
    nb = dense_block_rows(num_rows, world)
    if b_local.shape[0] != nb:
        raise ValueError(f"b_local must have {nb} rows (got {b_local.shape[0]})")
    b_local = b_local.contiguous()
    if out is None:
        out = torch.empty((world * nb, b_local.shape[1]), dtype=b_local.dtype, device=b_local.device)
    dist.all_gather_into_tensor(out, b_local, group=group)
    return out[:num_rows]


def _hip_spmm(reduce: str, rowptr, col, value, mat):
    from . import ops  # the HIP core; loads (or fails loudly) on first use

    return ops._spmm(reduce, rowptr, col, value, mat, want_arg=False)[0]  # `out` only: min/max skip arg_out


class RowPartitionedSpMM:
    """out_local = reduce-SpMM(A[r0:r1, :], all_gather(B_blocks)).

    local_spmm is the rank-local kernel, (reduce, rowptr, col, value, mat) ->
    out; it defaults to the HIP SpMM and exists as a parameter only so that the
    CPU/gloo tests can check the partitioning and the collective without a GPU.
    """

    def __init__(self, shard: RowShard, group=None, reduce: str = "sum",
                 local_spmm: Optional[Callable] = None):
        self.shard, self.group, self.reduce = shard, group, reduce
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.block_rows = dense_block_rows(shard.num_cols, self.world)
        self._local_spmm = local_spmm or _hip_spmm
        self._gather_buf: Optional[torch.Tensor] = None

    @classmethod
    def from_global(cls, rowptr, col, value, num_cols: int, group=None, reduce: str = "sum",
                    balance: str = "nnz", local_spmm: Optional[Callable] = None):
        """Every rank holds the whole CSR (e.g. loaded from disk) and keeps its block."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        M = rowptr.numel() - 1
        bounds = partition_rows_by_nnz(rowptr, world) if balance == "nnz" else partition_rows_evenly(M, world)
        return cls(shard_csr(rowptr, col, value, num_cols, bounds, rank), group, reduce, local_spmm)

    def local_dense_block(self, B: torch.Tensor) -> torch.Tensor:
        """This rank's (zero-padded) row block of a full B — a helper for
        callers that start from a replicated B."""
        nb, r = self.block_rows, self.rank
        blk = B[r * nb:(r + 1) * nb]
        if blk.shape[0] < nb:
            pad = torch.zeros((nb - blk.shape[0], B.shape[1]), dtype=B.dtype, device=B.device)
            blk = torch.cat([blk, pad])
        return blk.contiguous()

    def gather(self, b_local: torch.Tensor) -> torch.Tensor:
        shape = (self.world * self.block_rows, b_local.shape[1])
        buf = self._gather_buf
        if buf is None or buf.shape != shape or buf.dtype != b_local.dtype or buf.device != b_local.device:
            buf = self._gather_buf = torch.empty(shape, dtype=b_local.dtype, device=b_local.device)
        return all_gather_dense(b_local, self.shard.num_cols, self.group, out=buf)

    def __call__(self, b_local: torch.Tensor, feature_chunks: int = 1) -> torch.Tensor:
        """out_local [m_local, F].  feature_chunks = 1: one all-gather of B, then
        the local SpMM.  feature_chunks = C > 1 (opt-in): B is cut into C column
        slices; all C all-gathers are queued at once on the collective's own
        stream and the SpMM of slice c starts as soon as slice c has landed, so
        the kernel runs under the exchange of the later slices (every reduce
        is element-wise over columns, so slices are independent; the K <= 64
        kernels are as efficient per byte as the K = 128 one)."""
        s = self.shard
        if feature_chunks <= 1:
            return self._local_spmm(self.reduce, s.rowptr, s.col, s.value, self.gather(b_local))
        F = b_local.shape[1]
        bounds = [(F * c) // feature_chunks for c in range(feature_chunks + 1)]
        works, bufs = [], []
        for c in range(feature_chunks):
            part = b_local[:, bounds[c]:bounds[c + 1]].contiguous()
            buf = torch.empty((self.world * self.block_rows, part.shape[1]), dtype=part.dtype,
                              device=part.device)
            works.append(dist.all_gather_into_tensor(buf, part, group=self.group, async_op=True))
            bufs.append(buf)
        outs = []
        for c in range(feature_chunks):
            works[c].wait()  # the compute stream waits for slice c only
            outs.append(self._local_spmm(self.reduce, s.rowptr, s.col, s.value,
                                         bufs[c][:s.num_cols]))
        return torch.cat(outs, dim=1)

    def spmm_only(self, b_full: torch.Tensor) -> torch.Tensor:
        """Local kernel on an already-assembled B (B replicated / reused)."""
        s = self.shard
        return self._local_spmm(self.reduce, s.rowptr, s.col, s.value, b_full)


def gather_rows_to_root(out_local: torch.Tensor, bounds: List[int], group=None, dst: int = 0
                        ) -> Optional[torch.Tensor]:
    """Test/IO helper: concatenate every rank's output rows on rank `dst`."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    pad = max(sizes)
    buf = torch.zeros((pad, out_local.shape[1]), dtype=out_local.dtype, device=out_local.device)
    buf[:out_local.shape[0]] = out_local
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([g[:n] for g, n in zip(gathered, sizes)])
