"""paddle_sparse_amd — MI355X-native sparse-op layer behind the
paddle_sparse.SparseTensor / SparseStorage API (hot path: SpMM, COO coalesce,
ind2ptr/ptr2ind, transpose, row/column reductions; plus the ops either side
of it that run on the same kernels).

Importing the package loads the C-ABI HIP core and fails loudly when it is
missing, like the reference does for `paddle_sparse_ops`
(paddle_sparse/__init__.py:8-14).  There is no CPU or eager fallback.
"""
from __future__ import annotations

import importlib
import sys

__version__ = "0.1.0"


def _running_the_builder() -> bool:
    """`python -m paddle_sparse_amd.build` imports this package before build.py runs, on a
    checkout where the library is exactly what is about to be built: that one invocation
    gets the bare package (build.py needs nothing of it), every other import loads the
    core or fails."""
    argv = getattr(sys, "orig_argv", None) or []
    for flag, mod in zip(argv, argv[1:]):
        if flag == "-m":
            return mod == f"{__name__}.build"
    return False


if not _running_the_builder():
    from . import _lib

    _lib.load()  # ImportError with build instructions if the .so is absent

    from . import ops  # noqa: E402

    # The reference parses a CUDA version out of this op unless it answers -1
    # (paddle_sparse/__init__.py:17-32); a HIP core has no CUDA version to report.
    cuda_version = int(ops.sparse_cuda_version().item())
    if cuda_version != -1:
        raise ImportError("libpaddle_sparse_hip.so reports a CUDA version; it is not the HIP core")

    # Public surface, module by module (same names as paddle_sparse/__init__.py:34-84;
    # to/from_torch_sparse stand where to/from_paddle_sparse do, and spmm / matmul /
    # spspmm are the README's "later" entries).  Importing a module also attaches
    # its methods to SparseTensor.
    _PUBLIC = (
        ("storage", ("SparseStorage",)),
        ("tensor", ("SparseTensor",)),
        ("slicing", ("narrow", "__narrow_diag__", "select", "index_select", "index_select_nnz",
                     "masked_select", "masked_select_nnz")),
        ("sample", ("permute", "sample", "sample_adj")),
        ("add", ("add", "add_", "add_nnz", "add_nnz_")),
        ("mul", ("mul", "mul_", "mul_nnz", "mul_nnz_")),
        ("reduce", ("sum", "mean", "min", "max")),
        ("cat", ("cat",)),
        ("convert", ("to_torch_sparse", "from_torch_sparse", "to_paddle_sparse", "from_paddle_sparse", "to_scipy",
                     "from_scipy", "eye")),
        ("coalesce", ("coalesce",)),
        ("transpose", ("transpose", "t")),
        ("matmul", ("spmm", "matmul")),
        ("spspmm", ("spspmm",)),
    )

    __all__ = ["__version__"]
    for _module, _names in _PUBLIC:
        _m = importlib.import_module(f"{__name__}.{_module}")
        for _n in _names:
            globals()[_n] = getattr(_m, _n)
        __all__.extend(_names)
    del _module, _names, _m, _n
