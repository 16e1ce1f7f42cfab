"""Small host helpers (counterpart of paddle_sparse/utils.py)."""
from __future__ import annotations

from typing import Any, Optional, Tuple

import torch

from . import ops


def index_sort(inputs: torch.Tensor, max_value: Optional[int] = None,
               with_sorted_inputs: Optional[bool] = False, check: bool = True
               ) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """paddle_sparse/utils.py:14-23.  The reference falls back to
    `inputs.argsort()`; here it is the HIP radix sort, and `max_value` (the
    key bound the reference already passes) sets the number of passes.
    The reference's call sites (constructor sort, csr2csc) have no host read
    behind the sort, and an argsort cannot return a wrong order: the sort's fault
    word is read here (one 4-byte read) and HipCoreError raised if a look-back wait
    of a radix pass gave up.  check=False skips the read (stream capture)."""
    return ops.index_sort(inputs, max_value, bool(with_sorted_inputs), check=check)


def is_scalar(other: Any) -> bool:
    return isinstance(other, (int, float))


def is_pinned_tensor(x: torch.Tensor) -> bool:
    return (not x.is_cuda) and x.is_pinned()
