"""GPU: the C-ABI calls are stream-ordered and allocation-free (caller owns
outputs and workspaces), so a framework can capture them into a HIP graph
(cdna_hip_programming.md Guideline 9).  Capture SpMM forward/backward pieces
and the sort, replay on new data, compare with eager results."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_spmm_and_sort_replay_from_a_hip_graph():
    from paddle_sparse_amd import ops

    g = torch.Generator(device="cuda").manual_seed(0)
    M, nnz, K = 20_000, 200_000, 128
    row = torch.sort(torch.randint(0, M, (nnz,), generator=g, device="cuda"))[0]
    col = torch.randint(0, M, (nnz,), generator=g, device="cuda")
    val = torch.randn(nnz, generator=g, device="cuda")
    rowptr = ops.ind2ptr(row, M)
    B = torch.randn(M, K, generator=g, device="cuda")
    G = torch.randn(M, K, generator=g, device="cuda")
    keys = torch.randint(0, 1 << 40, (300_000,), generator=g, device="cuda")

    # warm up on a side stream (required before capture), then capture
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.spmm_sum(rowptr, col, val, B)
        ops.spmm_value_bw(None, rowptr, col, B, G)
        ops.index_sort(keys, 1 << 40, with_sorted_inputs=True)
    torch.cuda.current_stream().wait_stream(s)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.spmm_sum(rowptr, col, val, B)
        out_max, arg = ops.spmm_max(rowptr, col, val, B)
        gv = ops.spmm_value_bw(None, rowptr, col, B, G)
        srt, perm = ops.index_sort(keys, 1 << 40, with_sorted_inputs=True)

    for trial in range(3):  # new data in the captured input buffers, replay
        B.copy_(torch.randn(M, K, generator=g, device="cuda"))
        G.copy_(torch.randn(M, K, generator=g, device="cuda"))
        keys.copy_(torch.randint(0, 1 << 40, (300_000,), generator=g, device="cuda"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ops.spmm_sum(rowptr, col, val, B))
        e_max, e_arg = ops.spmm_max(rowptr, col, val, B)
        assert torch.equal(out_max, e_max) and torch.equal(arg, e_arg)
        assert torch.equal(gv, ops.spmm_value_bw(None, rowptr, col, B, G))
        ts, tp = torch.sort(keys, stable=True)
        assert torch.equal(srt, ts) and torch.equal(perm, tp)


def test_backward_passes_over_the_csc_view_replay_from_a_hip_graph():
    """The one-pass backward kernels (sum and max, both gradients) take their
    scratch from the caller and read no device value on the host, so a whole
    forward + backward step of a fixed graph replays from a HIP graph."""
    from paddle_sparse_amd import SparseStorage, ops

    g = torch.Generator(device="cuda").manual_seed(1)
    M, nnz, K = 30_000, 300_000, 64
    key = torch.unique(torch.randint(0, M * M, (nnz,), generator=g, device="cuda"))
    row, col = torch.div(key, M, rounding_mode="floor"), key % M
    val = torch.randn(row.numel(), generator=g, device="cuda")
    st = SparseStorage(row=row, col=col, value=val, sparse_sizes=(M, M), is_sorted=True, trust_data=True)
    rowptr, csr2csc, colptr = st.rowptr(), st.csr2csc(), st.colptr()
    row_csc, inv, tags = st._row_in_csc_order(), st.csc2csr(), st._csc_edge_tags()
    B = torch.randn(M, K, generator=g, device="cuda")
    G = torch.randn(M, K, generator=g, device="cuda")

    def step():
        out = ops.spmm_sum(rowptr, col, val, B)
        gv, gm = ops.spmm_sum_bw_csc(colptr, row_csc, csr2csc, val, B, G, True, csc2csr=inv)
        out_max, arg = ops.spmm_max(rowptr, col, val, B)
        gv2, gm2 = ops.spmm_minmax_bw_csc(rowptr, colptr, row_csc, csr2csc, tags, val, B, G, arg, csc2csr=inv)
        return out, gv, gm, out_max, gv2, gm2

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    for trial in range(3):
        B.copy_(torch.randn(M, K, generator=g, device="cuda"))
        G.copy_(torch.randn(M, K, generator=g, device="cuda"))
        graph.replay()
        torch.cuda.synchronize()
        for got, want in zip(captured, step()):
            assert torch.equal(got, want)


def test_half_width_step_with_long_columns_replays_from_a_hip_graph():
    """Round 4: the half-width passes over the CSC view with the long-column workspace (list pre-pass, the two-role
    launch, the combine) and the edge-range forward that leaves the row-local arg_out read nothing on the host either:
    a bf16 forward + backward step — sum and max, hub columns of 3 000 entries — replays from a HIP graph."""
    from paddle_sparse_amd import SparseStorage, ops

    g = torch.Generator(device="cuda").manual_seed(2)
    M, N, K = 20_000, 9_000, 64
    row = torch.randint(0, M, (150_000,), generator=g, device="cuda")
    col = torch.randint(0, N, (150_000,), generator=g, device="cuda")
    col[:3000], col[3000:3700] = 7, N - 1  # two long columns of the CSC view
    key = torch.unique(row * N + col)
    row, col = torch.div(key, N, rounding_mode="floor"), key % N
    val = torch.randn(row.numel(), generator=g, device="cuda")
    st = SparseStorage(row=row, col=col, value=val, sparse_sizes=(M, N), is_sorted=True, trust_data=True)
    rowptr, colptr, row_csc = st.rowptr(), st.colptr(), st._row_in_csc_order()
    assert st._csc_view()._longest_row() > 128
    w = ops.gather_rows(val, st.csr2csc())
    tags = st._csc_edge_tags(2)
    coo_row = st.row()
    B = torch.randn(N, K, generator=g, device="cuda").to(torch.bfloat16)
    G = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)

    def step():
        out = ops._spmm("sum", rowptr, col, val, B)[0]
        gv, gm = ops.spmm_half_sum_bw_csc(colptr, row_csc, w, B, G, True)
        out_max, _, words = ops._spmm("max", rowptr, col, val, B, want_arg=False, want_arg_bytes=2, row=coo_row,
                                      algo="edge_ranges")
        gv2, gm2 = ops.spmm_half_minmax_bw_csc(colptr, row_csc, tags, w, B, G, words)
        return out, gv, gm, out_max, words, gv2, gm2

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    for trial in range(3):
        B.copy_(torch.randn(N, K, generator=g, device="cuda").to(torch.bfloat16))
        G.copy_(torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16))
        graph.replay()
        torch.cuda.synchronize()
        for got, want in zip(captured, step()):
            assert torch.equal(got, want)
