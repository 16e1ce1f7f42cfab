#!/usr/bin/env python3
"""Row gathers (psa_gather_rows / psa_gather_rows_window): the halo exchange's send-side pack (1.43 M sorted rows of
512 B), the hub-row pack (65 536 rows) and a 128-byte column slice of the same rows; us and TB/s moved (read + write)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import event_ms
from paddle_sparse_amd import ops
dev = torch.device("cuda", 0)
B = torch.randn(2_000_000, 128, device=dev)
idx = torch.sort(torch.randperm(2_000_000, device=dev)[:1_430_000]).values
for name, fn, nb in (("gather_rows_window 1.43 M sorted rows x 512 B", lambda: ops.gather_rows_window(B, idx, 0, 128), 1.43e6 * 1024),
                     ("gather_rows 65 536 hub rows x 512 B", lambda: ops.gather_rows(B, idx[:65536]), 65536 * 1024),
                     ("gather_rows_window 1.43 M rows x 128 B slice", lambda: ops.gather_rows_window(B, idx, 32, 32), 1.43e6 * 256),
                     ("gather_rows 20 M floats through a permutation", None, None)):
    if fn is None: continue
    fn(); ms = event_ms(fn, 20)
    print(f"{name:55s} {ms*1e3:9.1f} us  {nb/ms/1e9:6.2f} TB/s")
