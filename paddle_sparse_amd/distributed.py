"""Row-partitioned multi-GPU SpMM: one process per GPU, RCCL over xGMI.

Not in the reference (it has no distributed code at all);
this is the north-star's multi-GPU path:

  * the sparse A is split into `world` contiguous row blocks, balanced by nnz
    through rowptr (each rank keeps its rebased rowptr slice and its col/value
    slice, and is the only writer of out[r0:r1, :] — no output exchange);
  * the dense B is row-sharded in equal blocks (rank r owns rows
    [r*nb, (r+1)*nb), nb = ceil(N / world)); every step reassembles what the
    rank needs of it in one of three ways —
      "full"      ONE `all_gather_into_tensor` of all of B (RCCL's all-gather),
      "full_p2p"  the same bytes as world - 1 direct peer copies posted together
                  (`batch_isend_irecv`: one send and one receive per peer inside one
                  RCCL group, every block one hop over its own xGMI link — the full
                  mesh has 7 links per GPU, a ring uses them differently),
      "halo"      only the rows of B the rank's column ids touch, ONE
                  `all_to_all_single` with split sizes, the local SpMM on the
                  compacted operand —
    and runs the local HIP SpMM with the block's own per-matrix plan
    (`RowShard.storage()`: kernel family from the row statistics, COO row ids,
    compact copy of the hub rows — what `SparseTensor.matmul` does on one GPU);
  * coalesce / index_sort / ind2ptr stay single-GPU (replicas only).

Steady state: exchange buffers, send buffers and (optionally) the output live
on the `RowPartitionedSpMM` object.  A buffer handed to a collective is
recorded on the collective's stream; allocated afresh every step it cannot go
back to the caching allocator before that stream's event completes, and a host
that runs ahead of the GPU then falls through to hipMalloc every step (round
2's committed rehearsal: 2.08 ms per step over 3 steps, 6.92 ms over 25).

Backward (`RowPartitionedSpMM.apply`): a differentiable exchange composed with the
ordinary differentiable local product.  grad of the local block of B = this rank's
rows of sum_r A_r^T grad_out_r: every rank runs its block's backward (the one-pass
kernels over the block's CSC view: every reduction, trained values, half width) and
ONE `reduce_scatter_tensor` (full) or the halo's all_to_all run backwards followed
by a segmented sum along the plan's return route (halo) brings the partial sums to their owners; the gradient
of a block's edge values stays on its rank.

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

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

EXCHANGES = ("full", "full_p2p", "halo")


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
    _storages: Dict[int, object] = field(default_factory=dict, repr=False, compare=False)

    @property
    def num_rows(self) -> int:
        return self.row_end - self.row_begin

    @property
    def nnz(self) -> int:
        return self.col.numel()

    def storage(self, col: Optional[torch.Tensor] = None, num_cols: Optional[int] = None):
        """The block as a SparseStorage (structure only, memoised per column array): the object
        that carries the per-matrix plan of the single-GPU product — `_spmm_algo()` from the
        block's own row statistics, `row()`, `_hot_columns()`, the CSC view for the backward.
        `col` / `num_cols`: the halo form's compacted column ids and operand height."""
        from .storage import SparseStorage

        col = self.col if col is None else col
        key = col.data_ptr()
        st = self._storages.get(key)
        if st is None:
            st = SparseStorage(rowptr=self.rowptr, col=col, value=None,
                               sparse_sizes=(self.num_rows, self.num_cols if num_cols is None else num_cols),
                               is_sorted=True, trust_data=True)
            self._storages[key] = st
        return st


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


class _Works:
    """Several asynchronous works waited for as one."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()


def peer_copy_dense(buf: torch.Tensor, part: torch.Tensor, group=None, async_op: bool = False):
    """The full exchange as direct peer copies: block r of `buf` ([world * nb, w]) receives rank
    r's `part` ([nb, w]); every rank posts one send and one receive per peer in ONE batch (an RCCL
    group: all world - 1 transfers run concurrently, each over the link to its peer) and copies
    its own block locally.  Same result as all_gather_into_tensor(buf, part)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nb = part.shape[0]
    buf[rank * nb:(rank + 1) * nb].copy_(part)
    ops = []
    for d in range(1, world):  # peer order rotated by rank: at every position of the batch the pairs are disjoint
        dst, src = (rank + d) % world, (rank - d) % world
        # dst / src are ranks OF `group`: the positional `peer` of P2POp is a GLOBAL rank (they differ in any
        # group other than WORLD), `group_peer` is the group's own numbering
        ops.append(dist.P2POp(dist.isend, part, group=group, group_peer=dst))
        ops.append(dist.P2POp(dist.irecv, buf[src * nb:(src + 1) * nb], group=group, group_peer=src))
    if not ops:
        return None
    works = _Works(dist.batch_isend_irecv(ops))
    if async_op:
        return works
    works.wait()
    return None


def _hip_spmm_planned(reduce: str, st, value, mat, out=None):
    from .matmul import spmm_planned  # the HIP core; loads (or fails loudly) on first use

    if out is not None and mat.dtype in (torch.float16, torch.bfloat16):
        # a half-width B travels as it is (half the bytes on the fabric) and runs the half-width kernels, which
        # allocate their result: one copy into the caller's buffer / column slice
        out.copy_(spmm_planned(st, value, mat, reduce))
        return out
    return spmm_planned(st, value, mat, reduce, out=out)


def _hip_pack(src, idx, col0, width, out=None):
    from . import ops

    return ops.gather_rows_window(src, idx, col0, width, out=out)


def _torch_pack(src, idx, col0, width, out=None):
    res = src[idx, col0:col0 + width]
    if out is None:
        return res.contiguous()
    out.copy_(res)
    return out


@dataclass
class HaloPlan:
    """Which rows of B this rank needs from every rank, and which of its own rows
    every rank needs from it — built once per matrix (structure only)."""
    need_counts: List[int]        # rows received from rank p per step
    send_counts: List[int]        # rows sent to rank q per step
    send_idx: torch.Tensor        # int64[sum(send_counts)]: LOCAL row ids of b_local, grouped by destination
    col_local: torch.Tensor       # int64[nnz_local]: col remapped into the received (compacted) B
    num_needed: int               # rows of the compacted B = distinct columns of the local block
    # the backward's way home, built at the first backward (`return_route`): the returned rows grouped by the local
    # row they belong to — a stable order of send_idx (requesters stay in rank order) and its CSR pointer
    back_order: Optional[torch.Tensor] = None   # int64[sum(send_counts)]
    back_ptr: Optional[torch.Tensor] = None     # int64[block_rows + 1]

    def return_route(self, block_rows: int) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.back_order is None:
            idx = self.send_idx
            if idx.is_cuda:
                from . import ops

                srt, order = ops.index_sort(idx, max_value=max(block_rows, 1), with_sorted_inputs=True, check=True)
                self.back_ptr = ops.ind2ptr(srt, block_rows)
            else:  # CPU / gloo tests
                srt, order = torch.sort(idx, stable=True)
                ptr = torch.zeros(block_rows + 1, dtype=torch.int64)
                ptr[1:] = torch.cumsum(torch.bincount(srt, minlength=block_rows)[:block_rows], 0)
                self.back_ptr = ptr
            self.back_order = order.contiguous()
        return self.back_order, self.back_ptr


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
    exchange = "full_p2p": the same bytes as direct peer copies (`peer_copy_dense`).
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

    Buffers: the exchange and send buffers of every slice shape are kept on the
    object (see the module docstring); `keep_output=True` keeps `out` there too —
    the returned tensor is then overwritten by the next call (a benchmark or an
    inference loop that consumes it at once; autograd callers leave it off or pass
    `out=`).

    local_spmm is the rank-local kernel, (reduce, rowptr, col, value, mat, out) ->
    out; it exists as a parameter only so that the CPU/gloo tests can check the
    partitioning and the collectives without a GPU.  The default is the HIP SpMM
    through the block's SparseStorage (`RowShard.storage()`), i.e. with the
    per-matrix plan; plan=False calls the kernel on the raw arrays (algo auto).
    """

    def __init__(self, shard: RowShard, group=None, reduce: str = "sum",
                 local_spmm: Optional[Callable] = None, exchange: str = "full", plan: bool = True,
                 keep_output: bool = False):
        if exchange not in EXCHANGES:
            raise ValueError(f"exchange must be one of {EXCHANGES}")
        self.shard, self.group, self.reduce, self.exchange = shard, group, reduce, exchange
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.block_rows = dense_block_rows(shard.num_cols, self.world)
        self._custom_spmm = local_spmm
        self.plan = plan and local_spmm is None
        self.keep_output = keep_output
        self._pack = _hip_pack if shard.col.is_cuda else _torch_pack
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self.halo: Optional[HaloPlan] = plan_halo(shard, self.block_rows, group) if exchange == "halo" else None

    @classmethod
    def from_global(cls, rowptr, col, value, num_cols: int, group=None, reduce: str = "sum",
                    balance: str = "nnz", local_spmm: Optional[Callable] = None, exchange: str = "full", **kw):
        """Every rank holds the whole CSR (e.g. loaded from disk) and keeps its block."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        M = rowptr.numel() - 1
        bounds = partition_rows_by_nnz(rowptr, world) if balance == "nnz" else partition_rows_evenly(M, world)
        return cls(shard_csr(rowptr, col, value, num_cols, bounds, rank), group, reduce, local_spmm, exchange, **kw)

    # ---- the rank-local kernel ----------------------------------------------------------
    def _operand_cols(self) -> Tuple[torch.Tensor, int]:
        """(column ids, operand height) of the local SpMM for this exchange."""
        if self.halo is None:
            return self.shard.col, self.shard.num_cols
        return self.halo.col_local, self.halo.num_needed

    def local_storage(self):
        col, height = self._operand_cols()
        return self.shard.storage(col, height)

    def _local(self, operand: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
        s = self.shard
        col, _ = self._operand_cols()
        if self._custom_spmm is not None:
            return self._custom_spmm(self.reduce, s.rowptr, col, s.value, operand, out)
        if self.plan:
            return _hip_spmm_planned(self.reduce, self.local_storage(), s.value, operand, out)
        from . import ops

        if out is not None and operand.dtype in (torch.float16, torch.bfloat16):
            # as in _hip_spmm_planned: the half-width kernels allocate their result, one copy into the caller's slice
            out.copy_(ops._spmm(self.reduce, s.rowptr, col, s.value, operand, want_arg=False)[0])
            return out
        return ops._spmm(self.reduce, s.rowptr, col, s.value, operand, want_arg=False, out=out)[0]

    # ---- buffers that live as long as the object ---------------------------------------
    def _buffer(self, kind: str, shape: Tuple[int, ...], like: torch.Tensor, slot: int = 0) -> torch.Tensor:
        key = (kind, slot, tuple(shape), like.dtype, like.device)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty(shape, dtype=like.dtype, device=like.device)
        return buf

    def buffer_bytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self._bufs.values())

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
        """The full exchange alone: all of B (one all-gather), in the object's buffer."""
        buf = self._buffer("recv", (self.world * self.block_rows, b_local.shape[1]), b_local)
        return all_gather_dense(b_local, self.shard.num_cols, self.group, out=buf)

    def _exchange_slice(self, b_local: torch.Tensor, c0: int, width: int, async_op: bool, slot: int = 0):
        """Start the exchange of columns [c0, c0 + width) of B; returns (work | None, operand)."""
        if b_local.shape[0] != self.block_rows:
            raise ValueError(f"b_local must have {self.block_rows} rows (got {b_local.shape[0]})")
        s, F = self.shard, b_local.shape[1]
        if self.halo is None:
            if width == F and b_local.is_contiguous():
                part = b_local
            else:  # the slice as a dense block, in a buffer of its own (the collective's stream reads it)
                part = self._buffer("part", (self.block_rows, width), b_local, slot)
                part.copy_(b_local[:, c0:c0 + width])
            buf = self._buffer("recv", (self.world * self.block_rows, width), b_local, slot)
            if self.exchange == "full_p2p":
                work = peer_copy_dense(buf, part, self.group, async_op=async_op)
            else:
                work = dist.all_gather_into_tensor(buf, part, group=self.group, async_op=async_op)
            return work, buf[:s.num_cols]
        h = self.halo
        send = self._buffer("send", (h.send_idx.numel(), width), b_local, slot)
        self._pack(b_local.contiguous(), h.send_idx, c0, width, send)
        recv = self._buffer("recv", (h.num_needed, width), b_local, slot)
        work = dist.all_to_all_single(recv, send, output_split_sizes=h.need_counts, input_split_sizes=h.send_counts,
                                      group=self.group, async_op=async_op)
        return work, recv

    def __call__(self, b_local: torch.Tensor, feature_chunks: int = 1, out: Optional[torch.Tensor] = None
                 ) -> torch.Tensor:
        """out_local [m_local, F]; see the class docstring for the knobs."""
        s, F = self.shard, b_local.shape[1]
        if out is None and self.keep_output:
            out = self._buffer("out", (s.num_rows, F), b_local)
        if feature_chunks <= 1:
            _, operand = self._exchange_slice(b_local, 0, F, async_op=False)
            return self._local(operand, out)
        bounds = [(F * c) // feature_chunks for c in range(feature_chunks + 1)]
        pending = [self._exchange_slice(b_local, bounds[c], bounds[c + 1] - bounds[c], async_op=True, slot=c)
                   for c in range(feature_chunks)]
        if out is None:
            out = torch.empty((s.num_rows, F), dtype=b_local.dtype, device=b_local.device)
        for c, (work, operand) in enumerate(pending):
            if work is not None:
                work.wait()  # the compute stream waits for slice c only
            self._local(operand, out[:, bounds[c]:bounds[c + 1]])
        return out

    def spmm_only(self, b_operand, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Local kernel(s) on an already-assembled operand: all of B (full exchanges) or the
        compacted rows (exchange "halo"), as `exchange_only` returns it — one tensor, or the
        list of column slices of the sliced form (one kernel per slice, written in place)."""
        slices = b_operand if isinstance(b_operand, (list, tuple)) else [b_operand]
        F = sum(t.shape[1] for t in slices)
        if out is None and self.keep_output:
            out = self._buffer("out", (self.shard.num_rows, F), slices[0])
        if len(slices) == 1:
            return self._local(slices[0], out)
        if out is None:
            out = torch.empty((self.shard.num_rows, F), dtype=slices[0].dtype, device=slices[0].device)
        c0 = 0
        for t in slices:
            self._local(t, out[:, c0:c0 + t.shape[1]])
            c0 += t.shape[1]
        return out

    def exchange_only(self, b_local: torch.Tensor, feature_chunks: int = 1):
        """The step's data movement alone (timing / reuse of B across SpMMs), slice after slice on
        the caller's stream: the assembled operand, or the list of its column slices."""
        F = b_local.shape[1]
        if feature_chunks <= 1:
            return self._exchange_slice(b_local, 0, F, async_op=False)[1]
        bounds = [(F * c) // feature_chunks for c in range(feature_chunks + 1)]
        return [self._exchange_slice(b_local, bounds[c], bounds[c + 1] - bounds[c], async_op=False, slot=c)[1]
                for c in range(feature_chunks)]

    # ---- autograd ------------------------------------------------------------------------
    def apply(self, b_local: torch.Tensor, value: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The step as a differentiable function of this rank's block of B — and, on the device, of the block's
        edge values.  Two autograd functions composed: a differentiable EXCHANGE (forward: the operand the rank's
        columns need; backward: the contributions to the rows of B go back to their owners — ONE
        reduce_scatter_tensor of the [world * nb, F] partial sums for the full exchanges, the halo's all-to-all
        run backwards plus an add by `send_idx` for the halo), then the ordinary differentiable local product
        (`SparseTensor.matmul` of the block: every reduction incl. min / max, trained values — their gradient
        stays on the rank that owns the rows —, half-width operands, the one-pass backward over the block's CSC
        view).  `value`: the block's edge values as a tensor that may require grad (default: the shard's own).
        With the CPU test hook (`local_spmm`) only sum / mean with fixed values differentiate."""
        if self._custom_spmm is not None or not b_local.is_cuda:
            if self.reduce not in ("sum", "mean"):
                raise NotImplementedError("RowPartitionedSpMM.apply with a local_spmm hook differentiates sum / mean")
            return _PartitionedSpMM.apply(b_local, self)
        from .matmul import spmm_sparse
        from .tensor import SparseTensor

        operand = _ExchangeB.apply(b_local, self)
        value = self.shard.value if value is None else value
        st = self.local_storage()  # ONE storage per block: the CSC view, tags and routes built by a backward stay for the next
        st.set_value_(value, layout="coo")
        return spmm_sparse(SparseTensor.from_storage(st), operand, self.reduce)

    def _exchange_fresh(self, b_local: torch.Tensor) -> torch.Tensor:
        """The exchange into buffers of its own (autograd keeps the operand until the backward: a second forward
        through the same object must not overwrite it)."""
        s, F = self.shard, b_local.shape[1]
        b_local = b_local.contiguous()
        if self.halo is None:
            buf = torch.empty((self.world * self.block_rows, F), dtype=b_local.dtype, device=b_local.device)
            if self.exchange == "full_p2p":
                peer_copy_dense(buf, b_local, self.group)
            else:
                dist.all_gather_into_tensor(buf, b_local, group=self.group)
            return buf[:s.num_cols]
        h = self.halo
        send = self._pack(b_local, h.send_idx, 0, F, None)
        recv = torch.empty((h.num_needed, F), dtype=b_local.dtype, device=b_local.device)
        dist.all_to_all_single(recv, send, output_split_sizes=h.need_counts, input_split_sizes=h.send_counts, group=self.group)
        return recv

    def _return_grad(self, part: torch.Tensor) -> torch.Tensor:
        """grad of the assembled operand ([N, F], or [num_needed, F] for the halo) -> grad of this rank's block of B."""
        F, nb = part.shape[1], self.block_rows
        if self.halo is None:
            if part.shape[0] != self.world * nb:  # pad to whole blocks
                padded = torch.zeros((self.world * nb, F), dtype=part.dtype, device=part.device)
                padded[:part.shape[0]] = part
                part = padded
            grad_local = torch.empty((nb, F), dtype=part.dtype, device=part.device)
            dist.reduce_scatter_tensor(grad_local, part.contiguous(), group=self.group)
            return grad_local
        h = self.halo
        back = torch.empty((h.send_idx.numel(), F), dtype=part.dtype, device=part.device)
        dist.all_to_all_single(back, part.contiguous(), output_split_sizes=h.send_counts, input_split_sizes=h.need_counts,
                               group=self.group)
        # a row may be asked for by several ranks: its copies are summed in requester order by a segmented
        # reduction over the route built once per plan (no atomics: two runs give the same bits, like every
        # single-GPU backward of the path)
        order, ptr = h.return_route(nb)
        if back.is_cuda:
            from . import ops

            acc = back if back.dtype == torch.float32 else back.float()
            return ops.segment_csr(acc, ptr, "sum", perm=order).to(part.dtype)
        acc = back[order].to(torch.float32)
        return torch.segment_reduce(acc, "sum", offsets=ptr, axis=0, initial=0.0).to(part.dtype)

    def _grad_operand(self, grad_out: torch.Tensor) -> torch.Tensor:
        """A_r^T grad_out_r as an [operand height, F] matrix (rank-local)."""
        s = self.shard
        col, height = self._operand_cols()
        if self._custom_spmm is not None or not grad_out.is_cuda:
            # CPU / test hook: the transpose as an explicit CSR (structure from torch, values gathered), same kernel hook
            deg = (s.rowptr[1:] - s.rowptr[:-1])
            row = torch.repeat_interleave(torch.arange(s.num_rows, dtype=torch.int64, device=col.device), deg)
            w = s.value if s.value is not None else torch.ones(col.numel(), dtype=grad_out.dtype, device=col.device)
            if self.reduce == "mean":
                w = w / deg.clamp(min=1).to(w.dtype)[row]
            order = torch.sort(col, stable=True)[1]
            colptr = torch.zeros(height + 1, dtype=torch.int64, device=col.device)
            colptr[1:] = torch.cumsum(torch.bincount(col, minlength=height), 0)
            fn = self._custom_spmm
            if fn is None:
                raise RuntimeError("RowPartitionedSpMM on CPU tensors needs the local_spmm hook")
            return fn("sum", colptr, row[order].contiguous(), w[order].contiguous(), grad_out.contiguous(), None)
        from .matmul import spmm_transposed_planned

        return spmm_transposed_planned(self.local_storage(), s.value, grad_out.contiguous(), self.reduce == "mean")

    def _backward(self, grad_out: torch.Tensor) -> torch.Tensor:
        return self._return_grad(self._grad_operand(grad_out))  # [N, F] (full) or [num_needed, F] (halo) -> the block's rows


class _ExchangeB(torch.autograd.Function):
    """The exchange of B as a differentiable function: forward = the assembled operand, backward = its gradient's
    rows brought to (and summed at) the ranks that own them."""

    @staticmethod
    def forward(ctx, b_local, op: RowPartitionedSpMM):
        ctx.op = op
        return op._exchange_fresh(b_local)

    @staticmethod
    def backward(ctx, grad_operand):
        return ctx.op._return_grad(grad_operand.contiguous()), None


class _PartitionedSpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b_local, op: RowPartitionedSpMM):
        ctx.op = op
        keep = op.keep_output
        op.keep_output = False  # autograd owns what it returns
        try:
            return op(b_local)
        finally:
            op.keep_output = keep

    @staticmethod
    def backward(ctx, grad_out):
        return ctx.op._backward(grad_out), None


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
