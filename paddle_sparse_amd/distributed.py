"""Row-partitioned multi-GPU SpMM: one process per GPU, RCCL over xGMI.

Not in the reference (it has no distributed code at all);
this is the north-star's multi-GPU path:

  * the sparse A is split into `world` contiguous row blocks, balanced by nnz
    through rowptr (each rank keeps its rebased rowptr slice and its col/value
    slice, and is the only writer of out[r0:r1, :] — no output exchange);
  * the dense B is row-sharded in equal blocks (rank r owns rows
    [r*nb, (r+1)*nb), nb = ceil(N / world)); every step either reassembles it
    with ONE all-gather (`torch.distributed.all_gather_into_tensor`, backend
    "nccl" = RCCL on ROCm) and runs the local HIP SpMM on the full B, or — the
    halo form — moves only the rows the rank's columns touch with ONE
    all_to_all_single and runs the SpMM on the compacted operand;
  * coalesce / index_sort / ind2ptr stay single-GPU (replicas only).

xGMI arithmetic that decides what this can reach (8-GPU full mesh, 7 links x
~153 GB/s per GPU): every rank must receive (world-1)/world of B every step;
at N = 16M, F = 128 that is 7.2 GB per rank, >= 6.7 ms even at the full
per-GPU ingest rate, against ~2 ms of local SpMM — the exchange, not the
kernel, bounds the step.  bench.py reports both.  Nothing here has run on
more than one GPU: the numbers above are arithmetic, the code is covered by
gloo world-2 tests on CPU and a world-1 RCCL test on the device.

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


def _hip_spmm(reduce: str, rowptr, col, value, mat, out=None):
    from . import ops  # the HIP core; loads (or fails loudly) on first use

    return ops._spmm(reduce, rowptr, col, value, mat, want_arg=False, out=out)[0]  # `out` only: min/max skip arg_out


def _hip_pack(src, idx, col0, width):
    from . import ops

    return ops.gather_rows_window(src, idx, col0, width)


def _torch_pack(src, idx, col0, width):
    return src[idx, col0:col0 + width].contiguous()


@dataclass
class HaloPlan:
    """Which rows of B this rank needs from every rank, and which of its own rows
    every rank needs from it — built once per matrix (structure only)."""
    need_counts: List[int]        # rows received from rank p per step
    send_counts: List[int]        # rows sent to rank q per step
    send_idx: torch.Tensor        # int64[sum(send_counts)]: LOCAL row ids of b_local, grouped by destination
    col_local: torch.Tensor       # int64[nnz_local]: col remapped into the received (compacted) B
    num_needed: int               # rows of the compacted B = distinct columns of the local block


def plan_halo(shard: RowShard, block_rows: int, group=None) -> HaloPlan:
    """Halo of a row block: the distinct columns of its entries.  Column ids are
    owned in equal blocks (rank p owns [p * block_rows, (p + 1) * block_rows)), so
    the sorted distinct ids fall apart into one contiguous run per source rank;
    the runs' lengths and the block-local ids go to the owners with two
    all-to-alls.  The local SpMM then runs on the received rows alone, with col
    replaced by its position among the distinct ids."""
    world = dist.get_world_size(group)
    col = shard.col
    uniq = torch.unique(col)  # sorted
    owner = torch.div(uniq, block_rows, rounding_mode="floor")
    need = torch.bincount(owner, minlength=world)[:world]
    asked = torch.empty_like(need)
    dist.all_to_all_single(asked, need, group=group)
    need_counts, send_counts = need.tolist(), asked.tolist()
    local_ids = (uniq - owner * block_rows).contiguous()
    send_idx = torch.empty(sum(send_counts), dtype=torch.int64, device=col.device)
    dist.all_to_all_single(send_idx, local_ids, output_split_sizes=send_counts, input_split_sizes=need_counts,
                           group=group)
    col_local = torch.searchsorted(uniq, col).contiguous()
    return HaloPlan(need_counts, send_counts, send_idx, col_local, int(uniq.numel()))


class RowPartitionedSpMM:
    """out_local = reduce-SpMM(A[r0:r1, :], B) with B row-sharded over the ranks.

    exchange = "full": every step reassembles all of B with one all-gather.
    exchange = "halo": every step moves only the rows of B that the rank's
    column ids touch (`plan_halo`, once per matrix): each rank packs the rows its
    peers asked for (one HIP gather), ONE all_to_all_single with split sizes
    moves them, and the local SpMM runs on the compacted operand with remapped
    column ids — the same edges in the same order, so the result equals the
    full exchange's bit for bit.  On a uniform random graph with 10 entries per
    row at 8 ranks that is 71 % of B per rank; graphs with locality need far less.

    feature_chunks = C > 1 overlaps exchange and compute: B is cut into C column
    slices, all C exchanges are queued up front on the collective's stream, and
    the SpMM of slice c runs as soon as slice c has landed, writing its columns of
    `out` in place (psa_spmm_coo's `ldo`: no concatenation; the send side packs
    rows and slice in one pass, psa_gather_rows_window).  Why slices of the
    feature dimension and not one block per source rank: on a full xGMI mesh all
    peers deliver at the same time over their own links, so per-source blocks
    all land together; and partial products per source would have to be summed
    into `out`, (world - 1) more read-modify-write passes over M x F floats —
    more HBM traffic than the SpMM itself at 8 ranks.

    local_spmm is the rank-local kernel, (reduce, rowptr, col, value, mat, out) ->
    out; it defaults to the HIP SpMM and exists as a parameter only so that the
    CPU/gloo tests can check the partitioning and the collectives without a GPU.
    """

    def __init__(self, shard: RowShard, group=None, reduce: str = "sum",
                 local_spmm: Optional[Callable] = None, exchange: str = "full"):
        if exchange not in ("full", "halo"):
            raise ValueError("exchange must be 'full' or 'halo'")
        self.shard, self.group, self.reduce, self.exchange = shard, group, reduce, exchange
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.block_rows = dense_block_rows(shard.num_cols, self.world)
        self._local_spmm = local_spmm or _hip_spmm
        self._pack = _hip_pack if shard.col.is_cuda else _torch_pack
        self._gather_buf: Optional[torch.Tensor] = None
        self.halo: Optional[HaloPlan] = plan_halo(shard, self.block_rows, group) if exchange == "halo" else None

    @classmethod
    def from_global(cls, rowptr, col, value, num_cols: int, group=None, reduce: str = "sum",
                    balance: str = "nnz", local_spmm: Optional[Callable] = None, exchange: str = "full"):
        """Every rank holds the whole CSR (e.g. loaded from disk) and keeps its block."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        M = rowptr.numel() - 1
        bounds = partition_rows_by_nnz(rowptr, world) if balance == "nnz" else partition_rows_evenly(M, world)
        return cls(shard_csr(rowptr, col, value, num_cols, bounds, rank), group, reduce, local_spmm, exchange)

    # ---- bytes on the fabric ----------------------------------------------------------
    def rows_received_per_step(self) -> int:
        """Rows of B that arrive from OTHER ranks every step."""
        if self.halo is not None:
            return sum(n for p, n in enumerate(self.halo.need_counts) if p != self.rank)
        return (self.world - 1) * self.block_rows

    def bytes_received_per_step(self, feat: int, itemsize: int = 4) -> int:
        return self.rows_received_per_step() * feat * itemsize

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
        """The full exchange alone: all of B (one all-gather)."""
        shape = (self.world * self.block_rows, b_local.shape[1])
        buf = self._gather_buf
        if buf is None or buf.shape != shape or buf.dtype != b_local.dtype or buf.device != b_local.device:
            buf = self._gather_buf = torch.empty(shape, dtype=b_local.dtype, device=b_local.device)
        return all_gather_dense(b_local, self.shard.num_cols, self.group, out=buf)

    def _exchange_slice(self, b_local: torch.Tensor, c0: int, width: int, async_op: bool):
        """Start the exchange of columns [c0, c0 + width) of B; returns (work | None, operand, col)."""
        if b_local.shape[0] != self.block_rows:
            raise ValueError(f"b_local must have {self.block_rows} rows (got {b_local.shape[0]})")
        s, F = self.shard, b_local.shape[1]
        if self.halo is None:
            part = b_local if width == F else b_local[:, c0:c0 + width].contiguous()
            buf = torch.empty((self.world * self.block_rows, width), dtype=b_local.dtype, device=b_local.device)
            work = dist.all_gather_into_tensor(buf, part.contiguous(), group=self.group, async_op=async_op)
            return work, buf[:s.num_cols], s.col
        h = self.halo
        send = self._pack(b_local.contiguous(), h.send_idx, c0, width)
        recv = torch.empty((h.num_needed, width), dtype=b_local.dtype, device=b_local.device)
        work = dist.all_to_all_single(recv, send, output_split_sizes=h.need_counts, input_split_sizes=h.send_counts,
                                      group=self.group, async_op=async_op)
        return work, recv, h.col_local

    def __call__(self, b_local: torch.Tensor, feature_chunks: int = 1) -> torch.Tensor:
        """out_local [m_local, F]; see the class docstring for the two knobs."""
        s, F = self.shard, b_local.shape[1]
        if feature_chunks <= 1:
            _, operand, col = self._exchange_slice(b_local, 0, F, async_op=False)
            return self._local_spmm(self.reduce, s.rowptr, col, s.value, operand, None)
        bounds = [(F * c) // feature_chunks for c in range(feature_chunks + 1)]
        pending = [self._exchange_slice(b_local, bounds[c], bounds[c + 1] - bounds[c], async_op=True)
                   for c in range(feature_chunks)]
        out = torch.empty((s.num_rows, F), dtype=b_local.dtype, device=b_local.device)
        for c, (work, operand, col) in enumerate(pending):
            work.wait()  # the compute stream waits for slice c only
            self._local_spmm(self.reduce, s.rowptr, col, s.value, operand, out[:, bounds[c]:bounds[c + 1]])
        return out

    def spmm_only(self, b_operand: torch.Tensor) -> torch.Tensor:
        """Local kernel on an already-assembled operand: all of B (exchange "full")
        or the compacted rows (exchange "halo", e.g. from `exchange_only`)."""
        s = self.shard
        col = s.col if self.halo is None else self.halo.col_local
        return self._local_spmm(self.reduce, s.rowptr, col, s.value, b_operand, None)

    def exchange_only(self, b_local: torch.Tensor) -> torch.Tensor:
        """The step's data movement alone (timing / reuse of B across SpMMs)."""
        return self._exchange_slice(b_local, 0, b_local.shape[1], async_op=False)[1]


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
