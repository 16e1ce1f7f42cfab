"""paddle_sparse_amd — MI355X-native sparse-op layer behind the
paddle_sparse.SparseTensor / SparseStorage API (hot path only: SpMM, COO
coalesce, ind2ptr/ptr2ind, transpose, row/column reductions).

Importing the package loads the C-ABI HIP core and fails loudly when it is
missing, like the reference does for `paddle_sparse_ops`
(paddle_sparse/__init__.py:8-14).  There is no CPU or eager fallback.
"""
from __future__ import annotations

__version__ = "0.1.0"

from . import _lib

_lib.load()  # ImportError with build instructions if the .so is absent

from . import ops  # noqa: E402

# paddle_sparse/__init__.py:17-32 parses a CUDA version unless the op
# returns -1; the HIP core always answers -1.
cuda_version = int(ops.sparse_cuda_version().item())
assert cuda_version == -1

from .storage import SparseStorage  # noqa: E402,F401
from .tensor import SparseTensor  # noqa: E402,F401
from .slicing import narrow, select, index_select, index_select_nnz  # noqa: E402,F401
from .slicing import masked_select, masked_select_nnz  # noqa: E402,F401
from .add import add, add_, add_nnz, add_nnz_  # noqa: E402,F401
from .mul import mul, mul_, mul_nnz, mul_nnz_  # noqa: E402,F401
from .reduce import sum, mean, min, max  # noqa: E402,F401,A004
from .convert import to_torch_sparse, from_torch_sparse  # noqa: E402,F401
from .convert import to_scipy, from_scipy  # noqa: E402,F401
from .coalesce import coalesce  # noqa: E402,F401
from .transpose import transpose, t  # noqa: E402,F401
from .matmul import spmm, matmul  # noqa: E402,F401
from .spspmm import spspmm  # noqa: E402,F401
from .cat import cat  # noqa: E402,F401
from .sample import sample, sample_adj, permute  # noqa: E402,F401

__all__ = [
    "SparseStorage",
    "SparseTensor",
    "narrow",
    "select",
    "index_select",
    "index_select_nnz",
    "masked_select",
    "masked_select_nnz",
    "add",
    "add_",
    "add_nnz",
    "add_nnz_",
    "mul",
    "mul_",
    "mul_nnz",
    "mul_nnz_",
    "sum",
    "mean",
    "min",
    "max",
    "to_torch_sparse",
    "from_torch_sparse",
    "to_scipy",
    "from_scipy",
    "coalesce",
    "transpose",
    "spmm",
    "matmul",
    "spspmm",
    "cat",
    "sample",
    "sample_adj",
    "permute",
    "__version__",
]
