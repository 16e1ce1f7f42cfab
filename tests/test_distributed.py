"""CPU suite, world_size 2 over gloo: the N > 1 path of the row-partitioned
SpMM — nnz-balanced row partition, CSR sharding, both exchanges of B (one
all-gather of everything; the halo form: one all_to_all_single of only the rows
a rank's columns touch, column ids remapped) and the in-place feature slices.
The rank-local kernel is the HIP SpMM in production; here (no GPU) the oracle
is plugged in AS THE CHECKER through the `local_spmm` test hook, so what is
verified is everything around the kernel.  A world-1 run on the device with
the real HIP kernel is at the end (marked gpu)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_local_spmm(reduce, rowptr, col, value, mat, out=None):
    import oracle

    res, _ = oracle.spmm(reduce, rowptr.numpy(), col.numpy(),
                         None if value is None else value.numpy(), mat.contiguous().numpy())
    res = torch.from_numpy(res)
    if out is None:
        return res
    out.copy_(res)  # a column slice of the caller's output matrix
    return out


def _worker(rank, world, port, M, N, nnz, F, reduce, balance, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from paddle_sparse_amd import distributed as pd
        from util import skewed_csr

        # every rank builds the same global problem from the seed
        row, rowptr, col, val = skewed_csr(M, N, seed=7, long_rows=(3, M // 2), long_deg=nnz)
        B = np.random.default_rng(1).standard_normal((N, F)).astype(np.float32)
        t_rowptr, t_col, t_val = torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val)

        op = pd.RowPartitionedSpMM.from_global(t_rowptr, t_col, t_val, N, reduce=reduce,
                                               balance=balance, local_spmm=_oracle_local_spmm)
        bounds = (pd.partition_rows_by_nnz(t_rowptr, world) if balance == "nnz"
                  else pd.partition_rows_evenly(M, world))
        s = op.shard
        assert (s.row_begin, s.row_end) == (bounds[rank], bounds[rank + 1])
        assert int(s.rowptr[0]) == 0 and int(s.rowptr[-1]) == s.nnz
        # the shard is exactly this rank's slice of the global CSR
        e0, e1 = rowptr[s.row_begin], rowptr[s.row_end]
        assert np.array_equal(s.col.numpy(), col[e0:e1]) and np.array_equal(s.value.numpy(), val[e0:e1])

        b_local = op.local_dense_block(torch.from_numpy(B))
        assert b_local.shape[0] == pd.dense_block_rows(N, world)
        b_full = op.gather(b_local)
        assert np.array_equal(b_full.numpy(), B)  # the collective reassembles B bit-exactly

        out_local = op(b_local)
        ref, _ = oracle.spmm(reduce, rowptr, col, val, B)
        assert np.array_equal(out_local.numpy(), ref[s.row_begin:s.row_end])
        # column-sliced form (all-gathers queued up front, SpMM per slice, written in place)
        out_sliced = op(b_local, feature_chunks=5)
        assert np.array_equal(out_sliced.numpy(), out_local.numpy())
        assert op.rows_received_per_step() == (world - 1) * pd.dense_block_rows(N, world)

        # halo exchange: only the rows of B this rank's columns touch travel; same bits
        halo = pd.RowPartitionedSpMM(s, reduce=reduce, local_spmm=_oracle_local_spmm, exchange="halo")
        out_halo = halo(b_local)
        assert np.array_equal(out_halo.numpy(), out_local.numpy())
        assert np.array_equal(halo(b_local, feature_chunks=3).numpy(), out_local.numpy())
        assert np.array_equal(halo.spmm_only(halo.exchange_only(b_local)).numpy(), out_local.numpy())
        nb = pd.dense_block_rows(N, world)
        mine = np.unique(col[e0:e1])
        h = halo.halo
        assert h.num_needed == mine.size and sum(h.need_counts) == mine.size
        assert h.need_counts == [int(((mine // nb) == p).sum()) for p in range(world)]
        assert halo.rows_received_per_step() == int(((mine // nb) != rank).sum())
        assert halo.bytes_received_per_step(F) == halo.rows_received_per_step() * F * 4
        assert halo.rows_received_per_step() <= op.rows_received_per_step()
        # what I send to q is what q needs from me: check through a gather of the counts
        counts = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor(h.need_counts))
        assert h.send_counts == [int(counts[q][rank]) for q in range(world)]
        assert np.array_equal(mine[h.col_local.numpy()], col[e0:e1])
        # direct peer copies (full_p2p): the same bytes as the all-gather, the same bits out
        p2p = pd.RowPartitionedSpMM(s, reduce=reduce, local_spmm=_oracle_local_spmm, exchange="full_p2p")
        assert np.array_equal(p2p.exchange_only(b_local).numpy(), B)
        assert np.array_equal(p2p(b_local).numpy(), out_local.numpy())
        assert np.array_equal(p2p(b_local, feature_chunks=4).numpy(), out_local.numpy())
        assert p2p.rows_received_per_step() == op.rows_received_per_step()

        # steady state: every buffer a collective touches lives on the object — after the first
        # step of a form no step allocates one again, and the second step's result is the first's
        for o, chunks in ((op, 1), (op, 5), (p2p, 4), (halo, 1), (halo, 3)):
            o(b_local, feature_chunks=chunks)
            held, ptrs = o.buffer_bytes(), sorted(t.data_ptr() for t in o._bufs.values())
            again = o(b_local, feature_chunks=chunks)
            assert o.buffer_bytes() == held and sorted(t.data_ptr() for t in o._bufs.values()) == ptrs
            assert np.array_equal(again.numpy(), out_local.numpy())
        # keep_output: the result lives on the object too and is overwritten by the next call
        kept = pd.RowPartitionedSpMM(s, reduce=reduce, local_spmm=_oracle_local_spmm, exchange="full", keep_output=True)
        first = kept(b_local)
        second = kept(b_local * 2)
        assert first.data_ptr() == second.data_ptr() and np.array_equal(second.numpy(), kept(b_local * 2).numpy())
        mine_out = torch.empty_like(out_local)
        assert kept(b_local, out=mine_out).data_ptr() == mine_out.data_ptr()
        assert np.array_equal(mine_out.numpy(), out_local.numpy())

        # backward through the step (sum / mean): grad of this rank's block of B = its rows of A^T G
        if reduce in ("sum", "mean"):
            G = np.random.default_rng(9).standard_normal((M, F)).astype(np.float32)
            w = val / np.maximum(np.diff(rowptr), 1)[row] if reduce == "mean" else val
            want = np.zeros((world * nb, F), np.float64)
            np.add.at(want, col, w[:, None].astype(np.float64) * G[row].astype(np.float64))
            for o in (op, p2p, halo):
                bl = b_local.clone().requires_grad_(True)
                o.apply(bl).backward(torch.from_numpy(G[s.row_begin:s.row_end]))
                np.testing.assert_allclose(bl.grad.numpy(), want[rank * nb:(rank + 1) * nb], rtol=1e-4, atol=1e-4)
        else:
            with pytest.raises(NotImplementedError):
                op.apply(b_local)
        full = pd.gather_rows_to_root(out_local, bounds)
        if rank == 0:
            assert np.array_equal(full.numpy(), ref)
            share = [int(rowptr[bounds[r + 1]] - rowptr[bounds[r]]) for r in range(world)]
            Path(result_path).write_text(",".join(map(str, share)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("reduce,balance,N", [("sum", "nnz", 401), ("mean", "rows", 400), ("max", "nnz", 257)])
def test_row_partitioned_spmm_world2_gloo(tmp_path, reduce, balance, N):
    import oracle

    oracle.build()
    world, M, F = 2, 300, 24
    result = tmp_path / "share.txt"
    mp.spawn(_worker, args=(world, _free_port(), M, N, 500, F, reduce, balance, str(result)),
             nprocs=world, join=True)
    share = list(map(int, result.read_text().split(",")))
    assert sum(share) > 0
    if balance == "nnz":  # two 500-edge rows dominate: nnz balance must split them apart
        assert max(share) <= 0.75 * sum(share)


def _worker_strong_split(rank, world, port, M, N, F):
    """bench.py --scaling strong in miniature: every rank builds the ONE matrix from the same seed, keeps its block of
    rows (balanced by nnz), owns an equal block of B; every exchange gives the single-process result, bit for bit,
    also when a rank's block is EMPTY (more ranks than rows with entries) and when N is not a multiple of the world."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from paddle_sparse_amd import distributed as pd

        rng = np.random.default_rng(17)
        deg = rng.integers(0, 7, M)
        if M <= world:  # fewer rows than ranks: some blocks are empty
            deg[:] = 3
        deg[M // 3] = 40  # one heavy row
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        row = np.repeat(np.arange(M), deg)
        col = rng.integers(0, N, rowptr[-1]).astype(np.int64)
        val = rng.standard_normal(rowptr[-1]).astype(np.float32)
        B = rng.standard_normal((N, F)).astype(np.float32)
        G = rng.standard_normal((M, F)).astype(np.float32)
        t = (torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val))
        bounds = pd.partition_rows_by_nnz(t[0], world)
        assert bounds[0] == 0 and bounds[-1] == M and all(a <= b for a, b in zip(bounds, bounds[1:]))
        shard = pd.shard_csr(*t, N, bounds, rank)
        ref, _ = oracle.spmm("sum", rowptr, col, val, B)
        nb = pd.dense_block_rows(N, world)
        want_g = np.zeros((world * nb, F), np.float64)
        np.add.at(want_g, col, val[:, None].astype(np.float64) * G[row].astype(np.float64))
        outs = []
        for exchange in pd.EXCHANGES:
            op = pd.RowPartitionedSpMM(shard, exchange=exchange, local_spmm=_oracle_local_spmm)
            b_local = op.local_dense_block(torch.from_numpy(B))
            assert b_local.shape == (nb, F)
            for chunks in (1, 3):
                out = op(b_local, feature_chunks=chunks)
                assert out.shape == (shard.num_rows, F)
                assert np.array_equal(out.numpy(), ref[shard.row_begin:shard.row_end]), (exchange, chunks)
            outs.append(out)
            bl = b_local.clone().requires_grad_(True)
            op.apply(bl).backward(torch.from_numpy(G[shard.row_begin:shard.row_end]))
            np.testing.assert_allclose(bl.grad.numpy(), want_g[rank * nb:(rank + 1) * nb], rtol=1e-4, atol=1e-4)
        full = pd.gather_rows_to_root(outs[0], bounds)
        if rank == 0:
            assert np.array_equal(full.numpy(), ref)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M,N", [(4, 90, 64), (4, 3, 10), (3, 50, 31), (8, 400, 203)])  # 8: the rank count of the real job
def test_strong_split_every_exchange_more_ranks_gloo(world, M, N):
    import oracle

    oracle.build()
    mp.spawn(_worker_strong_split, args=(world, _free_port(), M, N, 12), nprocs=world, join=True)


def _worker_halo_backward_bits(rank, world, port):
    """The halo form's backward sums the copies of a row of B that several ranks asked for along a route built once
    per plan (stable order of send_idx + its CSR pointer, a segmented sum) — no atomics: two runs give the same bits,
    the route is built once, and the gradient equals the full form's within 1e-5 * sum |terms|."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from paddle_sparse_amd import distributed as pd

        rng = np.random.default_rng(23)
        M, N, F = 240, 90, 20
        deg = rng.integers(2, 9, M)
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        row = np.repeat(np.arange(M), deg)
        col = rng.integers(0, N, rowptr[-1]).astype(np.int64)
        col[rng.random(col.size) < 0.3] = rng.integers(0, 5, 1)  # hub columns: asked for by every rank
        val = rng.standard_normal(col.size).astype(np.float32)
        B = rng.standard_normal((N, F)).astype(np.float32)
        G = rng.standard_normal((M, F)).astype(np.float32)
        args = (torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), N)
        full = pd.RowPartitionedSpMM.from_global(*args, local_spmm=_oracle_local_spmm)
        halo = pd.RowPartitionedSpMM.from_global(*args, local_spmm=_oracle_local_spmm, exchange="halo")
        s = halo.shard
        g_local = torch.from_numpy(G[s.row_begin:s.row_end])
        b_local = full.local_dense_block(torch.from_numpy(B))
        grads = []
        for o in (halo, halo, full):
            bl = b_local.clone().requires_grad_(True)
            o.apply(bl).backward(g_local)
            grads.append(bl.grad.numpy().copy())
        assert np.array_equal(grads[0], grads[1])  # bit-identical run to run
        order, ptr = halo.halo.return_route(halo.block_rows)
        assert halo.halo.return_route(halo.block_rows)[0] is order  # built once
        sent = halo.halo.send_idx.numpy()
        assert np.array_equal(sent[order.numpy()], np.sort(sent, kind="stable"))
        assert np.array_equal(ptr.numpy(), np.searchsorted(np.sort(sent), np.arange(halo.block_rows + 1)))
        nb = halo.block_rows
        abs_terms = np.zeros((world * nb, F), np.float64)
        np.add.at(abs_terms, col, np.abs(val[:, None].astype(np.float64) * G[row].astype(np.float64)))
        tol = 1e-5 * abs_terms[rank * nb:(rank + 1) * nb] + 1e-30
        assert np.all(np.abs(grads[0].astype(np.float64) - grads[2].astype(np.float64)) <= tol)
    finally:
        dist.destroy_process_group()


def test_halo_backward_is_bitwise_reproducible_world3_gloo():
    import oracle

    oracle.build()
    mp.spawn(_worker_halo_backward_bits, args=(3, _free_port()), nprocs=3, join=True)


def _worker_subgroup_p2p(rank, world, port):
    """peer_copy_dense inside a group whose ranks are NOT the global ranks (global 1, 2 -> group 0, 1): the P2P
    ops must name their peers in the group's numbering (the positional `peer` is a global rank)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from paddle_sparse_amd import distributed as pd

        members = [1, 2]
        grp = dist.new_group(members)  # every rank takes part in the creation
        if rank in members:
            g_rank = dist.get_rank(grp)
            assert g_rank == members.index(rank) and g_rank != rank
            nb, w = 5, 3
            part = torch.full((nb, w), float(10 + rank))
            buf = torch.zeros(2 * nb, w)
            pd.peer_copy_dense(buf, part, grp)
            want = torch.cat([torch.full((nb, w), float(10 + m)) for m in members])
            assert torch.equal(buf, want)
            buf2 = torch.zeros(2 * nb, w)
            work = pd.peer_copy_dense(buf2, part, grp, async_op=True)
            work.wait()
            assert torch.equal(buf2, want)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_peer_copies_inside_a_subgroup_use_group_ranks_gloo():
    mp.spawn(_worker_subgroup_p2p, args=(3, _free_port()), nprocs=3, join=True)


def _worker_skewed_halo(rank, world, port, result_path):
    """Banded graph with a few hub columns: most of a rank's columns are its own
    rows' neighbours, so the halo is a small fraction of B; results identical."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from paddle_sparse_amd import distributed as pd

        rng = np.random.default_rng(5)
        M = N = 600
        F = 16
        row = np.repeat(np.arange(M), 8)
        col = (row + rng.integers(-20, 21, row.size)) % N          # a band around the diagonal
        col[rng.random(row.size) < 0.05] = rng.integers(0, 4, 1)    # ... plus hub columns 0-3
        order = np.lexsort((col, row))
        row, col = row[order], col[order]
        val = rng.standard_normal(row.size).astype(np.float32)
        rowptr = np.searchsorted(row, np.arange(M + 1)).astype(np.int64)
        B = rng.standard_normal((N, F)).astype(np.float32)
        args = (torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), N)
        full = pd.RowPartitionedSpMM.from_global(*args, local_spmm=_oracle_local_spmm)
        halo = pd.RowPartitionedSpMM.from_global(*args, local_spmm=_oracle_local_spmm, exchange="halo")
        b_local = full.local_dense_block(torch.from_numpy(B))
        a, b = full(b_local), halo(b_local)
        assert np.array_equal(a.numpy(), b.numpy())
        ref, _ = oracle.spmm("sum", rowptr, col, val, B)
        assert np.array_equal(a.numpy(), ref[full.shard.row_begin:full.shard.row_end])
        if rank == 0:
            Path(result_path).write_text(f"{halo.rows_received_per_step()},{full.rows_received_per_step()}")
    finally:
        dist.destroy_process_group()


def test_halo_exchange_moves_a_fraction_of_b_on_a_graph_with_locality(tmp_path):
    import oracle

    oracle.build()
    result = tmp_path / "rows.txt"
    mp.spawn(_worker_skewed_halo, args=(2, _free_port(), str(result)), nprocs=2, join=True)
    halo_rows, full_rows = map(int, result.read_text().split(","))
    assert full_rows == 300 and halo_rows < 0.25 * full_rows  # the band's overhang (wrap-around) and the hubs


@pytest.mark.gpu
def test_row_partitioned_spmm_world1_on_the_device():
    """The multi-GPU step with the real HIP kernel as the local op and RCCL as the backend,
    at the only world size one GPU allows: full and halo exchange, with and without the
    in-place feature slices, against the single-GPU call."""
    from paddle_sparse_amd import distributed as pd
    from paddle_sparse_amd import ops
    from util import skewed_csr

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        M, N, F = 5000, 3000, 128
        row, rowptr, col, val = skewed_csr(M, N, seed=3, long_rows=(7, 2500), long_deg=4000)
        B = torch.from_numpy(np.random.default_rng(2).standard_normal((N, F)).astype(np.float32)).cuda()
        t_rowptr, t_col, t_val = (torch.from_numpy(x).cuda() for x in (rowptr, col, val))
        from paddle_sparse_amd import SparseTensor

        whole = SparseTensor(rowptr=t_rowptr, col=t_col, value=t_val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
        for reduce in ("sum", "mean", "max"):
            with torch.no_grad():  # the single-GPU product: the same per-matrix plan the rank-local step takes
                want = whole.matmul(B, reduce)
            for exchange in ("full", "full_p2p", "halo"):
                op = pd.RowPartitionedSpMM.from_global(t_rowptr, t_col, t_val, N, reduce=reduce, exchange=exchange)
                b_local = op.local_dense_block(B)
                assert torch.equal(op(b_local), want), (reduce, exchange)
                assert torch.equal(op(b_local), want)  # second step: the object's buffers again
                sliced = op(b_local, feature_chunks=4)  # K = 32 kernels: another summation order for sum / mean
                if reduce == "max":
                    assert torch.equal(sliced, want)
                else:
                    S = ops._spmm("sum", t_rowptr, t_col, t_val.abs(), B.abs())[0]
                    assert bool(((sliced - want).abs() <= 1e-5 * S + 1e-30).all())
                assert torch.equal(op.spmm_only(op.exchange_only(b_local)), want)
            assert op.halo.num_needed == int(torch.unique(t_col).numel()) and op.rows_received_per_step() == 0

        # the rank-local step takes the block's per-matrix plan, like SparseTensor.matmul on one GPU:
        # a power-law block (most rows empty or tiny, hub columns) runs the edge-range kernels on a
        # compact copy of the hub rows; same bits as the tensor surface, which makes the same choices
        rng = np.random.default_rng(11)
        M2, N2, F2 = 40_000, 400_000, 64
        deg = np.where(rng.random(M2) < 0.7, 0, rng.integers(1, 3, M2))
        deg[::500] = 3000
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        hubs = rng.integers(0, N2, 50_000)  # the 50 000 ids drawn here take 60 % of the entries
        c = np.where(rng.random(rp[-1]) < 0.6, hubs[rng.integers(0, hubs.size, rp[-1])], rng.integers(0, N2, rp[-1])).astype(np.int64)
        v = rng.standard_normal(rp[-1]).astype(np.float32)
        t_rp, t_c, t_v = (torch.from_numpy(x).cuda() for x in (rp, c, v))
        B2 = torch.randn(N2, F2, device="cuda")
        G2 = torch.randn(M2, F2, device="cuda")
        tensor = SparseTensor(rowptr=t_rp, col=t_c, value=t_v, sparse_sizes=(M2, N2), is_sorted=True, trust_data=True)
        for exchange in ("full", "halo"):
            op = pd.RowPartitionedSpMM.from_global(t_rp, t_c, t_v, N2, exchange=exchange)
            st = op.local_storage()
            assert st._spmm_algo() == "edge_ranges" and tensor.storage._spmm_algo() == "edge_ranges"
            if exchange == "full":
                assert st._hot_columns() is not None
            b_local = op.local_dense_block(B2)
            Bg = B2.clone().requires_grad_(True)
            want = tensor.matmul(Bg)
            want.backward(G2)
            assert torch.equal(op(b_local), want.detach())
            raw = pd.RowPartitionedSpMM.from_global(t_rp, t_c, t_v, N2, exchange=exchange, plan=False)
            S = ops._spmm("sum", t_rp, t_c, t_v.abs(), B2.abs())[0]
            assert bool(((raw(b_local) - want.detach()).abs() <= 1e-5 * S + 1e-30).all())
            # plan=False with a half-width operand and an output the object keeps / column slices — the
            # half-width kernels take no `out`, the raw branch must copy like the planned one
            raw_kept = pd.RowPartitionedSpMM.from_global(t_rp, t_c, t_v, N2, exchange=exchange, plan=False, keep_output=True)
            bh = b_local.to(torch.bfloat16)
            want_h = tensor.matmul(B2.to(torch.bfloat16)).float()
            for chunks in (1, 2):
                got_h = raw_kept(bh, feature_chunks=chunks)
                assert got_h.dtype == torch.bfloat16
                assert bool(((got_h.float() - want_h).abs() <= 2.0 ** -7 * want_h.abs() + 1e-5 * S + 1e-30).all())
            # backward through the step: reduce_scatter (full) / the halo run backwards
            bl = b_local.clone().requires_grad_(True)
            op.apply(bl).backward(G2)
            # |A|^T |G| bounds the rounding of either summation order
            bound = 1e-5 * SparseTensor(rowptr=t_rp, col=t_c, value=t_v.abs(), sparse_sizes=(M2, N2), is_sorted=True,
                                         trust_data=True).t().matmul(G2.abs()) + 1e-30
            assert bool(((bl.grad[:N2] - Bg.grad).abs() <= bound).all())
    finally:
        dist.destroy_process_group()


def _worker_device_gloo(rank, world, port, reduce):
    """Several ranks SHARING cuda:0 over gloo (it moves device tensors): the multi-rank logic with the real HIP
    kernels underneath — nnz-balanced blocks, the per-block plan, the HIP pack of the halo exchange, in-place
    feature slices, the backward through the step.  RCCL itself needs one GPU per rank and is not in this test."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from paddle_sparse_amd import SparseTensor, ops
        from paddle_sparse_amd import distributed as pd

        rng = np.random.default_rng(23)
        M, N, F = 30_000, 20_011, 64  # N not a multiple of the world: the last block of B is padded
        deg = np.where(rng.random(M) < 0.5, 0, rng.integers(1, 9, M))
        deg[::700] = 900  # hub rows: some blocks take the edge-range family, rows above 128 entries
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        nnz = int(rowptr[-1])
        hubs = rng.integers(0, N, 300)
        col = np.where(rng.random(nnz) < 0.4, hubs[rng.integers(0, 300, nnz)], rng.integers(0, N, nnz)).astype(np.int64)
        val = rng.standard_normal(nnz).astype(np.float32)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        t_rowptr, t_col, t_val = d(rowptr), d(col), d(val)
        B = d(rng.standard_normal((N, F)).astype(np.float32))
        G = d(rng.standard_normal((M, F)).astype(np.float32))
        whole = SparseTensor(rowptr=t_rowptr, col=t_col, value=t_val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
        Bg = B.clone().requires_grad_(True)
        vg = t_val.clone().requires_grad_(True)
        whole = SparseTensor(rowptr=t_rowptr, col=t_col, value=vg, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
        ref = whole.matmul(Bg, reduce)
        S = ops._spmm("sum", t_rowptr, t_col, t_val.abs(), B.abs())[0] + 1e-30
        ref.backward(G)
        whole = SparseTensor(rowptr=t_rowptr, col=t_col, value=t_val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
        Sg = SparseTensor(rowptr=t_rowptr, col=t_col, value=t_val.abs(), sparse_sizes=(M, N), is_sorted=True,
                          trust_data=True).t().matmul(G.abs()) + 1e-30  # sum |terms| of either gradient's sums
        Sv = ops.spmm_value_bw(None, t_rowptr, t_col, B.abs(), G.abs(), "sum") + 1e-30
        bounds = pd.partition_rows_by_nnz(t_rowptr, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        nb = pd.dense_block_rows(N, world)
        for exchange in pd.EXCHANGES:
            op = pd.RowPartitionedSpMM.from_global(t_rowptr, t_col, t_val, N, reduce=reduce, exchange=exchange)
            assert (op.shard.row_begin, op.shard.row_end) == (r0, r1)
            b_local = op.local_dense_block(B)
            for chunks in (1, 4):
                out = op(b_local, feature_chunks=chunks)
                if reduce == "max":
                    assert torch.equal(out, ref.detach()[r0:r1]), (exchange, chunks)
                else:
                    # a block may take another kernel family than the whole matrix: 1e-5 of the sum of |terms| (for mean: an over-estimate)
                    assert bool(((out - ref.detach()[r0:r1]).abs() <= 1e-5 * S[r0:r1]).all()), (exchange, chunks)
            held = op.buffer_bytes()
            op(b_local, feature_chunks=4)
            assert op.buffer_bytes() == held  # steady state: nothing new is allocated for the collectives
            # a bf16 B is exchanged as bf16 (half the bytes) and multiplied by the half-width kernels
            bh = b_local.to(torch.bfloat16)
            want_h = whole.matmul(B.to(torch.bfloat16), reduce)[r0:r1].float()
            for chunks in (1, 2):
                got_h = op(bh, feature_chunks=chunks)
                assert got_h.dtype == torch.bfloat16
                assert bool(((got_h.float() - want_h).abs() <= 1e-5 * S[r0:r1] + 2.0 ** -7 * want_h.abs()).all()), (exchange, chunks)
            # backward through the step: a differentiable exchange + the block's own autograd — every reduction,
            # and trained edge values (their gradient stays on the rank that owns the rows)
            bl = b_local.clone().requires_grad_(True)
            vl = op.shard.value.clone().requires_grad_(True)
            op.apply(bl, value=vl).backward(G[r0:r1])
            rows = min(nb, N - rank * nb)
            got, want = bl.grad[:rows], Bg.grad[rank * nb:rank * nb + rows]
            assert bool(((got - want).abs() <= 1e-5 * Sg[rank * nb:rank * nb + rows]).all()), exchange
            assert bool((bl.grad[rows:] == 0).all())  # the padding rows of the last block receive nothing
            e0, e1 = int(rowptr[r0]), int(rowptr[r1])
            assert bool(((vl.grad - vg.grad[e0:e1]).abs() <= 1e-5 * Sv[e0:e1]).all()), exchange
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,reduce", [(2, "sum"), (3, "mean"), (2, "max")])
def test_row_partitioned_spmm_more_ranks_on_one_device_over_gloo(world, reduce):
    mp.spawn(_worker_device_gloo, args=(world, _free_port(), reduce), nprocs=world, join=True)


def test_partition_helpers():
    from paddle_sparse_amd import distributed as pd

    rowptr = torch.tensor([0, 0, 10, 10, 11, 12, 40, 40])
    for world in (1, 2, 3, 8):
        b = pd.partition_rows_by_nnz(rowptr, world)
        assert b[0] == 0 and b[-1] == 7 and len(b) == world + 1
        assert all(b[i] <= b[i + 1] for i in range(world))
    assert pd.partition_rows_by_nnz(torch.tensor([0]), 4) == [0, 0, 0, 0, 0]  # empty matrix
    assert pd.partition_rows_evenly(10, 4) == [0, 2, 5, 7, 10]
    assert pd.dense_block_rows(10, 4) == 3 and pd.dense_block_rows(8, 4) == 2
