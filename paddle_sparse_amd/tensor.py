"""SparseTensor — thin façade over SparseStorage (paddle_sparse/tensor.py).

Same public surface as the reference class for the parts that sit on the hot
path (constructors, coo/csr/csc, value handling, coalesce, caches, size/stat
helpers, to_symmetric, dense / scipy / torch.sparse conversions).  Slicing
(`__getitem__`, narrow/select/index_select/masked_select) is attached by
slicing.py, add/mul by add.py/mul.py, reductions by reduce.py, SpMM by
matmul.py — the reference monkey-patches its methods the same way.
"""
from __future__ import annotations

from textwrap import indent
from typing import List, Optional, Tuple, Union

import numpy as np
import scipy.sparse
import torch

from . import ops
from .storage import SparseStorage, get_layout

# to_symmetric: above this many keys the two sorted halves are merged, below they are sorted
_MERGE_ABOVE = 1 << 16


class SparseTensor(object):
    storage: SparseStorage

    def __init__(
        self,
        row: Optional[torch.Tensor] = None,
        rowptr: Optional[torch.Tensor] = None,
        col: Optional[torch.Tensor] = None,
        value: Optional[torch.Tensor] = None,
        sparse_sizes: Optional[Tuple[Optional[int], Optional[int]]] = None,
        is_sorted: bool = False,
        trust_data: bool = False,
    ):
        self.storage = SparseStorage(row=row, rowptr=rowptr, col=col, value=value,
                                     sparse_sizes=sparse_sizes, is_sorted=is_sorted,
                                     trust_data=trust_data)

    # ---- constructors (tensor.py:48-213) -------------------------------------
    @classmethod
    def from_storage(cls, storage: SparseStorage):
        out = cls.__new__(cls)
        out.storage = storage
        return out

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor] = None,
                        sparse_sizes=None, is_sorted: bool = False, trust_data: bool = False):
        return cls(row=edge_index[0], col=edge_index[1], value=edge_attr,
                   sparse_sizes=sparse_sizes, is_sorted=is_sorted, trust_data=trust_data)

    @classmethod
    def from_dense(cls, mat: torch.Tensor, has_value: bool = True):
        if mat.dim() > 2:
            index = mat.abs().sum(list(range(2, mat.dim()))).nonzero()
        else:
            index = mat.nonzero()
        row, col = index[:, 0].contiguous(), index[:, 1].contiguous()
        value = mat[row, col] if has_value else None
        return cls(row=row, col=col, value=value, sparse_sizes=(mat.shape[0], mat.shape[1]),
                   is_sorted=True, trust_data=True)

    @classmethod
    def from_torch_sparse_coo_tensor(cls, mat: torch.Tensor, has_value: bool = True):
        """Counterpart of from_paddle_sparse_coo_tensor (tensor.py:109-131)."""
        mat = mat.coalesce()
        index = mat.indices()
        return cls(row=index[0].contiguous(), col=index[1].contiguous(),
                   value=mat.values() if has_value else None,
                   sparse_sizes=(mat.shape[0], mat.shape[1]), is_sorted=True, trust_data=True)

    @classmethod
    def from_torch_sparse_csr_tensor(cls, mat: torch.Tensor, has_value: bool = True):
        """Counterpart of from_paddle_sparse_csr_tensor (tensor.py:133-154)."""
        return cls(rowptr=mat.crow_indices().to(torch.int64), col=mat.col_indices().to(torch.int64),
                   value=mat.values() if has_value else None,
                   sparse_sizes=(mat.shape[0], mat.shape[1]), is_sorted=True, trust_data=True)

    @classmethod
    def eye(cls, M: int, N: Optional[int] = None, has_value: bool = True,
            dtype: Optional[torch.dtype] = None, device=None, fill_cache: bool = False):
        """tensor.py:156-213."""
        N = M if N is None else N
        k = min(M, N)
        row = torch.arange(k, dtype=torch.int64, device=device)
        rowptr = torch.arange(M + 1, dtype=torch.int64, device=device).clamp_(max=k)
        value = torch.ones(k, dtype=dtype, device=device) if has_value else None
        out = cls(row=row, rowptr=rowptr, col=row, value=value, sparse_sizes=(M, N),
                  is_sorted=True, trust_data=True)
        if fill_cache:
            st = out.storage
            st._rowcount = (torch.arange(M, device=device) < k).to(torch.int64)
            st._colptr = torch.arange(N + 1, dtype=torch.int64, device=device).clamp_(max=k)
            st._colcount = (torch.arange(N, device=device) < k).to(torch.int64)
            st._csr2csc = st._csc2csr = row
        return out

    def copy(self):
        return self.from_storage(self.storage.copy())

    def clone(self):
        return self.from_storage(self.storage.clone())

    def type(self, dtype: torch.dtype, non_blocking: bool = False):
        value = self.storage.value()
        if value is None or dtype == value.dtype:
            return self
        return self.from_storage(self.storage.type(dtype=dtype, non_blocking=non_blocking))

    def type_as(self, tensor: torch.Tensor, non_blocking: bool = False):
        return self.type(dtype=tensor.dtype, non_blocking=non_blocking)

    def to_device(self, device, non_blocking: bool = False):
        if torch.device(device) == self.device():
            return self
        return self.from_storage(self.storage.to_device(device, non_blocking=non_blocking))

    def device_as(self, tensor: torch.Tensor, non_blocking: bool = False):
        return self.to_device(device=tensor.device, non_blocking=non_blocking)

    # ---- formats (tensor.py:246-257) --------------------------------------------
    def coo(self):
        return self.storage.row(), self.storage.col(), self.storage.value()

    def csr(self):
        return self.storage.rowptr(), self.storage.col(), self.storage.value()

    def csc(self):
        self.storage.csr2csc()  # first: it leaves colptr and row[csr2csc] behind
        return self.storage.colptr(), self.storage._row_in_csc_order(), self.storage._value_in_csc_order()

    # ---- storage inheritance -------------------------------------------------------
    def has_value(self) -> bool:
        return self.storage.has_value()

    def set_value_(self, value: Optional[torch.Tensor], layout: Optional[str] = None):
        self.storage.set_value_(value, layout)
        return self

    def set_value(self, value: Optional[torch.Tensor], layout: Optional[str] = None):
        return self.from_storage(self.storage.set_value(value, layout))

    def sparse_sizes(self) -> Tuple[int, int]:
        return self.storage.sparse_sizes()

    def sparse_size(self, dim: int) -> int:
        return self.storage.sparse_sizes()[dim]

    def sparse_resize(self, sparse_sizes: Tuple[int, int]):
        return self.from_storage(self.storage.sparse_resize(sparse_sizes))

    def sparse_reshape(self, num_rows: int, num_cols: int):
        return self.from_storage(self.storage.sparse_reshape(num_rows, num_cols))

    def is_coalesced(self) -> bool:
        return self.storage.is_coalesced()

    def coalesce(self, reduce: str = "sum"):
        return self.from_storage(self.storage.coalesce(reduce))

    def fill_cache_(self):
        self.storage.fill_cache_()
        return self

    def clear_cache_(self):
        self.storage.clear_cache_()
        return self

    def __eq__(self, other) -> bool:
        if not isinstance(other, self.__class__) or self.sizes() != other.sizes():
            return False
        rowptr_a, col_a, value_a = self.csr()
        rowptr_b, col_b, value_b = other.csr()
        if (value_a is None) != (value_b is None):
            return False
        if not torch.equal(rowptr_a, rowptr_b) or not torch.equal(col_a, col_b):
            return False
        return value_a is None or torch.equal(value_a, value_b)

    __hash__ = object.__hash__

    # ---- utility ---------------------------------------------------------------------
    def fill_value_(self, fill_value: float, dtype: Optional[torch.dtype] = None):
        value = torch.full((self.nnz(),), fill_value, dtype=dtype, device=self.device())
        return self.set_value_(value, layout="coo")

    def fill_value(self, fill_value: float, dtype: Optional[torch.dtype] = None):
        value = torch.full((self.nnz(),), fill_value, dtype=dtype, device=self.device())
        return self.set_value(value, layout="coo")

    def sizes(self) -> List[int]:
        value = self.storage.value()
        tail = list(value.shape[1:]) if value is not None else []
        return list(self.sparse_sizes()) + tail

    def size(self, dim: int) -> int:
        return self.sizes()[dim]

    def dim(self) -> int:
        return len(self.sizes())

    def nnz(self) -> int:
        return self.storage.col().numel()

    def numel(self) -> int:
        value = self.storage.value()
        return value.numel() if value is not None else self.nnz()

    def density(self) -> float:
        if self.sparse_size(0) == 0 or self.sparse_size(1) == 0:
            return 0.0
        return self.nnz() / (self.sparse_size(0) * self.sparse_size(1))

    def sparsity(self) -> float:
        return 1 - self.density()

    def avg_row_length(self) -> float:
        return self.nnz() / self.sparse_size(0)

    def avg_col_length(self) -> float:
        return self.nnz() / self.sparse_size(1)

    def bandwidth(self) -> int:
        row, col, _ = self.coo()
        return int((row - col).abs_().max())

    def avg_bandwidth(self) -> float:
        row, col, _ = self.coo()
        return float((row - col).abs_().to(torch.float32).mean())

    def bandwidth_proportion(self, bandwidth: int) -> float:
        row, col, _ = self.coo()
        return int(((row - col).abs_() <= bandwidth).sum()) / self.nnz()

    def is_quadratic(self) -> bool:
        return self.sparse_size(0) == self.sparse_size(1)

    def is_symmetric(self) -> bool:
        if not self.is_quadratic():
            return False
        rowptr, col, value1 = self.csr()
        colptr, row, value2 = self.csc()
        if not torch.equal(rowptr, colptr) or not torch.equal(col, row):
            return False
        return value1 is None or value2 is None or bool((value1 == value2).all())

    def to_symmetric(self, reduce: str = "sum"):
        """tensor.py:415-451: union of A and A^T, duplicates reduced.  Built on
        the same sort + run-length + segmented-reduce kernels as coalesce()."""
        N = max(self.size(0), self.size(1))
        row, col, value = self.coo()
        n = row.numel()
        if 2 * n > _MERGE_ABOVE:
            # A is in (row, col) order and its CSC view is A^T in (row, col)
            # order: merge the two sorted streams instead of sorting 2n keys
            from .coalesce import _coalesce_two_sorted

            colptr, row_csc, value_csc = self.csc()
            new_row, new_col, value = _coalesce_two_sorted(row, col, value, ops.ptr2ind(colptr, n), row_csc,
                                                          value_csc, N, reduce)
            return SparseTensor(row=new_row, col=new_col, value=value, sparse_sizes=(N, N),
                                is_sorted=True, trust_data=True)
        both_r, both_c = torch.cat([row, col]), torch.cat([col, row])
        keys, _ = ops.make_keys(both_r, both_c, N)
        sorted_keys, perm, scratch = ops.index_sort(keys, N * N, with_sorted_inputs=True, keep_scratch=True)
        _, ptr, new_row, new_col = ops.unique_sorted(sorted_keys, N, after=scratch)
        if value is not None:
            # value of entry i (0 <= i < 2n) is value[i mod n]; fold it into perm
            src = torch.where(perm >= n, perm - n, perm)
            value = ops.segment_csr(value, ptr, reduce, perm=src)
        return SparseTensor(row=new_row, col=new_col, value=value, sparse_sizes=(N, N),
                            is_sorted=True, trust_data=True)

    def detach_(self):
        value = self.storage.value()
        if value is not None:
            value.detach_()
        return self

    def detach(self):
        value = self.storage.value()
        return self.set_value(value.detach() if value is not None else None, layout="coo")

    def requires_grad(self) -> bool:
        value = self.storage.value()
        return value.requires_grad if value is not None else False

    def requires_grad_(self, requires_grad: bool = True, dtype: Optional[torch.dtype] = None):
        if requires_grad and not self.has_value():
            self.fill_value_(1.0, dtype)
        value = self.storage.value()
        if value is not None:
            value.requires_grad_(requires_grad)
        return self

    def pin_memory(self):
        return self.from_storage(self.storage.pin_memory())

    def is_pinned(self) -> bool:
        return self.storage.is_pinned()

    def share_memory_(self):
        self.storage.share_memory_()
        return self

    def is_shared(self) -> bool:
        return self.storage.is_shared()

    def device(self):
        return self.storage.col().device

    def cpu(self):
        return self.to_device("cpu")

    def cuda(self, device: Optional[Union[int, str]] = None, non_blocking: bool = False):
        if device is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        elif isinstance(device, int):
            dev = torch.device("cuda", device)
        else:
            dev = torch.device(device)
        return self.to_device(dev, non_blocking)

    def is_cuda(self) -> bool:
        return self.storage.col().is_cuda

    def dtype(self):
        value = self.storage.value()
        return value.dtype if value is not None else torch.float32

    def is_floating_point(self) -> bool:
        value = self.storage.value()
        return torch.is_floating_point(value) if value is not None else True

    def bfloat16(self):
        return self.type(torch.bfloat16)

    def bool(self):
        return self.type(torch.bool)

    def byte(self):
        return self.type(torch.uint8)

    def char(self):
        return self.type(torch.int8)

    def half(self):
        return self.type(torch.float16)

    def float(self):
        return self.type(torch.float32)

    def double(self):
        return self.type(torch.float64)

    def short(self):
        return self.type(torch.int16)

    def int(self):
        return self.type(torch.int32)

    def long(self):
        return self.type(torch.int64)

    def to(self, *args, **kwargs):
        """tensor.py:606-689, reduced to what torch's own `.to` parser gives:
        a device and/or a dtype (or a tensor to take both from)."""
        if not args and not kwargs:
            raise TypeError("to() needs a device, a dtype or a tensor")
        other = kwargs.pop("other", None)
        if args and isinstance(args[0], torch.Tensor):
            other, args = args[0], args[1:]
        non_blocking = bool(kwargs.pop("non_blocking", False))
        if other is not None:
            device, dtype = other.device, other.dtype
        else:
            device, dtype, _, _ = torch._C._nn._parse_to(*args, **kwargs)
        out = self
        if dtype is not None:
            out = out.type(dtype=dtype, non_blocking=non_blocking)
        if device is not None:
            out = out.to_device(device=device, non_blocking=non_blocking)
        return out

    # ---- conversions (tensor.py:545-600, 796-856) ---------------------------------------
    def to_dense(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        row, col, value = self.coo()
        if value is not None:
            mat = torch.zeros(self.sizes(), dtype=value.dtype, device=self.device())
            mat[row, col] = value
        else:
            mat = torch.zeros(self.sizes(), dtype=dtype, device=self.device())
            mat[row, col] = 1
        return mat

    def to_torch_sparse_coo_tensor(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        row, col, value = self.coo()
        if value is None:
            value = torch.ones(self.nnz(), dtype=dtype, device=self.device())
        return torch.sparse_coo_tensor(torch.stack([row, col]), value, self.sizes())

    def to_torch_sparse_csr_tensor(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        rowptr, col, value = self.csr()
        if value is None:
            value = torch.ones(self.nnz(), dtype=dtype, device=self.device())
        return torch.sparse_csr_tensor(rowptr, col, value, self.sizes())

    def to_torch_sparse_csc_tensor(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """The reference's to_paddle_sparse_csc_tensor raises (Paddle has no CSC
        layout, tensor.py:587-591); torch has one, so the CSC view is handed over."""
        colptr, row, value = self.csc()
        if value is None:
            value = torch.ones(self.nnz(), dtype=dtype, device=self.device())
        return torch.sparse_csc_tensor(colptr, row, value, self.sizes())

    @classmethod
    def from_scipy(cls, mat, has_value: bool = True, device=None):
        colptr = None
        if isinstance(mat, scipy.sparse.csc_matrix):
            colptr = torch.from_numpy(mat.indptr.astype(np.int64)).to(device)
        csr = mat.tocsr()
        rowptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(device)
        coo = csr.tocoo()
        row = torch.from_numpy(coo.row.astype(np.int64)).to(device)
        col = torch.from_numpy(coo.col.astype(np.int64)).to(device)
        value = torch.from_numpy(coo.data).to(device) if has_value else None
        storage = SparseStorage(row=row, rowptr=rowptr, col=col, value=value,
                                sparse_sizes=tuple(coo.shape[:2]), colptr=colptr, is_sorted=True)
        return cls.from_storage(storage)

    def to_scipy(self, layout: Optional[str] = None, dtype: Optional[torch.dtype] = None):
        assert self.dim() == 2
        layout = get_layout(layout)
        host = lambda t: t.detach().cpu().numpy()
        ones = None if self.has_value() else torch.ones(self.nnz(), dtype=dtype).numpy()
        if layout == "coo":
            row, col, value = self.coo()
            data = host(value) if value is not None else ones
            return scipy.sparse.coo_matrix((data, (host(row), host(col))), self.sizes())
        if layout == "csr":
            rowptr, col, value = self.csr()
            data = host(value) if value is not None else ones
            return scipy.sparse.csr_matrix((data, host(col), host(rowptr)), self.sizes())
        colptr, row, value = self.csc()
        data = host(value) if value is not None else ones
        return scipy.sparse.csc_matrix((data, host(row), host(colptr)), self.sizes())

    def __repr__(self) -> str:
        i = " " * 6
        row, col, value = self.coo()
        infos = [f"row={indent(row.__repr__(), i)[len(i):]}",
                 f"col={indent(col.__repr__(), i)[len(i):]}"]
        if value is not None:
            infos.append(f"val={indent(value.__repr__(), i)[len(i):]}")
        infos.append(f"size={tuple(self.sizes())}, nnz={self.nnz()}, "
                     f"density={100 * self.density():.02f}%")
        i = " " * (len(self.__class__.__name__) + 1)
        body = indent(",\n".join(infos), i)[len(i):]
        return f"{self.__class__.__name__}({body})"
