"""Is a hipMemsetAsync issued by the core captured into a torch HIP graph?
psa_ind2ptr(numel=0) is a pure hipMemsetAsync of `out` (safe: no kernels)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from paddle_sparse_amd import _lib, ops

lib = _lib.load()
empty = torch.empty(0, dtype=torch.int64, device="cuda")
out = torch.ones(1025, dtype=torch.int64, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ops.ind2ptr(empty, 8)
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    _lib.check(lib.psa_ind2ptr(None, 0, 1024, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("after capture (memset must NOT have run yet): sum =", int(out.sum()))
for trial in range(3):
    out.fill_(1)
    graph.replay()
    torch.cuda.synchronize()
    print("replay", trial, "sum =", int(out.sum()), "(0 = memset replayed)")
