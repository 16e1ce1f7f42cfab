"""CPU suite: the C-ABI library loads and exports every symbol that
include/paddle_sparse_hip.h declares (no compute calls: no GPU needed)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "paddle_sparse_hip.h"


def declared_functions():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psa_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    names = declared_functions()
    assert "psa_spmm" in names and "psa_ind2ptr" in names and len(names) >= 7


def test_library_exports_every_declared_symbol():
    from paddle_sparse_amd import _lib

    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_binding_table_matches_header():
    from paddle_sparse_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.load()  # sets restype/argtypes for every symbol


def test_version_probes_without_gpu():
    from paddle_sparse_amd import _lib, ops

    lib = _lib.load()
    assert lib.psa_abi_version() == 1
    # csrc/version.cpp:14-22 / __init__.py:18-32: HIP build must report -1
    assert lib.psa_sparse_cuda_version() == -1
    v = ops.sparse_cuda_version()
    assert v.tolist() == [-1] and not v.is_cuda


def test_ops_reject_cpu_tensors():
    import torch
    from paddle_sparse_amd import ops

    ind = torch.tensor([0, 1, 1], dtype=torch.int64)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.ind2ptr(ind, 3)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.ptr2ind(ind, 3)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.spmm_sum(ind, ind, None, torch.zeros(2, 2))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from paddle_sparse_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_convert_round_trips_on_the_host():
    """test/test_convert.py:9-33 (no kernels involved): scipy and framework
    sparse tensors in and out of the (index, value) form."""
    import torch
    from paddle_sparse_amd import eye, from_scipy, from_torch_sparse, to_scipy, to_torch_sparse

    index = torch.tensor([[0, 0, 1, 2, 2], [0, 2, 1, 0, 1]])
    value = torch.tensor([1, 2, 4, 1, 3])
    out = from_scipy(to_scipy(index, value, 3, 3))
    assert out[0].tolist() == index.tolist() and out[1].tolist() == value.tolist()
    out = from_torch_sparse(to_torch_sparse(index, value, 3, 3).coalesce())
    assert out[0].tolist() == index.tolist() and out[1].tolist() == value.tolist()
    import paddle_sparse_amd as psa

    # the reference's names for the host framework's COO type (convert.py:9-14)
    assert psa.to_paddle_sparse is to_torch_sparse and psa.from_paddle_sparse is from_torch_sparse
    idx, val = eye(4, dtype=torch.float32)
    assert idx.tolist() == [[0, 1, 2, 3]] * 2 and val.tolist() == [1.0] * 4
    with pytest.raises(ValueError):
        to_scipy(index[0], value, 3, 3)
