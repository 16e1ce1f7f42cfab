#!/usr/bin/env python3
"""XCD mixing of the row blocks in the fp32 row-wave FORWARD (variant 28), against the default
order: uniform config 3 and R-MAT 21 as generated, spmm_sum and spmm_max (out + arg_out)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import event_ms, make_workload, rmat_graph  # noqa: E402
from paddle_sparse_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
F = 128
M = 2_000_000
graphs = {"uniform C3": (M,) + make_workload(M, M, 20_000_000, F, 0, dev)}
N, rowptr, row, col, val = rmat_graph(21, 20_000_000, dev)
graphs["R-MAT 21 as generated"] = (N, rowptr, col, val)
for name, (n, rp, c, v) in graphs.items():
    B = torch.randn(n, F, device=dev)
    for op in ("spmm_sum", "spmm_max"):
        fn = getattr(ops, op)
        line = f"{name}: {op} row waves:"
        ref = None
        for variant, tag in ((0, "blocks x mod 8 -> XCD x"), (28, "mixed"), (0, "again"), (28, "mixed again")):
            prev = lib.psa_spmm_set_variant(variant)
            try:
                out = fn(rp, c, v, B, algo="row_waves")
                out = out[0] if isinstance(out, tuple) else out
                if ref is None:
                    ref = out
                assert torch.equal(out, ref)
                line += f"  {tag} {event_ms(lambda: fn(rp, c, v, B, algo='row_waves'), 30):.4f} ms"
            finally:
                lib.psa_spmm_set_variant(prev)
        print(line, flush=True)
