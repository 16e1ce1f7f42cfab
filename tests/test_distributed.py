"""CPU suite, world_size 2 over gloo: the N > 1 path of the row-partitioned
SpMM — nnz-balanced row partition, CSR sharding, the all-gather of B and the
concatenation of per-rank outputs.  The rank-local kernel is the HIP SpMM in
production; here (no GPU) the oracle is plugged in AS THE CHECKER through the
`local_spmm` test hook, so what is verified is everything around the kernel."""
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


def _oracle_local_spmm(reduce, rowptr, col, value, mat):
    import oracle

    out, _ = oracle.spmm(reduce, rowptr.numpy(), col.numpy(),
                         None if value is None else value.numpy(), mat.numpy())
    return torch.from_numpy(out)


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
        # column-sliced form (all-gathers queued up front, SpMM per slice)
        out_sliced = op(b_local, feature_chunks=5)
        assert np.array_equal(out_sliced.numpy(), out_local.numpy())
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
