"""Where the 10k-entry coalesce spends its time on the host: each piece of
ops.coalesce_chain / coalesce() timed alone (perf_counter, 2000 repeats)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

import torch

import paddle_sparse_amd as ps
from paddle_sparse_amd import _lib, ops


def timed(fn, reps=2000, sync=True):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    n, M, N = 10_000, 1000, 1000
    index = torch.stack([torch.randint(0, M, (n,), generator=g, device="cuda"),
                         torch.randint(0, N, (n,), generator=g, device="cuda")])
    value = torch.randn(n, generator=g, device="cuda")
    row, col = index[0].contiguous(), index[1].contiguous()
    lib = _lib.load()
    buf = torch.empty(3 * n + 2, dtype=torch.int64, device="cuda")
    status = buf[3 * n:]
    out = buf[2 * n:3 * n].view(torch.float32)[:n]
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        lib.psa_coalesce_small_fused(row.data_ptr(), col.data_ptr(), value.data_ptr(), 0, n, M, N, 0,
                                     buf.data_ptr(), out.data_ptr(), status.data_ptr(), st)

    print(f"ps.coalesce(index, value)            {timed(lambda: ps.coalesce(index, value, M, N)):7.1f} us")
    print(f"ops.coalesce_chain(row, col, value)  {timed(lambda: ops.coalesce_chain(row, col, value, M, N)):7.1f} us")
    print(f"launch only, back to back (GPU time) {timed(launch):7.1f} us")

    def launch_read():
        launch()
        return status.tolist()
    print(f"launch + status.tolist()             {timed(launch_read):7.1f} us")

    def launch_sync():
        launch()
        torch.cuda.current_stream().synchronize()
    print(f"launch + stream.synchronize()        {timed(launch_sync):7.1f} us")
    print(f"status.tolist() alone (idle stream)  {timed(lambda: status.tolist()):7.1f} us")
    print(f"stream.synchronize() alone           {timed(lambda: torch.cuda.current_stream().synchronize()):7.1f} us")
    print(f"index[0].contiguous() x2             {timed(lambda: (index[0].contiguous(), index[1].contiguous())):7.1f} us")
    print(f"torch.empty(3n+2) + 3 views          {timed(lambda: (lambda b: (b[3*n:], b[2*n:3*n].view(torch.float32)[:n]))(torch.empty(3*n+2, dtype=torch.int64, device='cuda'))):7.1f} us")
    print(f"ops._index x2 + _gpu                 {timed(lambda: (ops._index(row, 'r'), ops._index(col, 'c'), ops._gpu(value, 'v'))):7.1f} us")
    print(f"current_stream().cuda_stream         {timed(lambda: torch.cuda.current_stream().cuda_stream):7.1f} us")
    print(f"current_device()                     {timed(lambda: torch.cuda.current_device()):7.1f} us")
    print(f"6 x data_ptr()                       {timed(lambda: (row.data_ptr(), col.data_ptr(), value.data_ptr(), buf.data_ptr(), out.data_ptr(), status.data_ptr())):7.1f} us")
    print(f"_stack_index + 2 result views        {timed(lambda: (buf[:2*9000].view(2, 9000), out[:9000])):7.1f} us")
    host = torch.empty(2, dtype=torch.int64, pin_memory=True)
    hp = host.data_ptr()

    def launch_pinned():
        lib.psa_coalesce_small_fused(row.data_ptr(), col.data_ptr(), value.data_ptr(), 0, n, M, N, 0,
                                     buf.data_ptr(), out.data_ptr(), hp, st)
        torch.cuda.current_stream().synchronize()
        return host.tolist()
    try:
        t = timed(launch_pinned)
        ref = launch_read()
        print(f"launch, status in pinned host memory {t:7.1f} us   same words: {launch_pinned() == ref}")
    except Exception as e:  # noqa: BLE001
        print("pinned status failed:", e)


if __name__ == "__main__":
    main()
