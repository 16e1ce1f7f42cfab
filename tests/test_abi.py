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


# ---- the Paddle custom-op shim (integration/paddle_shim) against the header ----------------------
SHIM = ROOT / "integration" / "paddle_shim" / "paddle_sparse_hip_ops.cc"
INTEGRATION = ROOT / "INTEGRATION.md"


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _top_level_args(text, open_paren):
    """Number of top-level arguments of the call whose '(' is at open_paren."""
    depth, args, i, seen = 0, 1, open_paren, False
    while True:
        c = text[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return args if seen else 0
        elif c == "," and depth == 1:
            args += 1
        elif depth >= 1 and not c.isspace():
            seen = True
        i += 1


def header_arity():
    text = _strip_comments(HEADER.read_text())
    out = {}
    for m in re.finditer(r"\b(psa_[a-z0-9_]+)\s*\(", text):
        n = _top_level_args(text, m.end() - 1)
        inner = text[m.end():text.index(")", m.end())].strip()
        out[m.group(1)] = 0 if inner == "void" else n
    return out


def test_shim_calls_match_the_header_arity():
    arity = header_arity()
    text = _strip_comments(SHIM.read_text())
    calls = [(m.group(1), _top_level_args(text, m.end() - 1)) for m in re.finditer(r"\b(psa_[a-z0-9_]+)\s*\(", text)]
    assert len(calls) > 40
    for name, n in calls:
        assert name in arity, f"shim calls {name}, which the header does not declare"
        assert n == arity[name], f"{name}: shim passes {n} arguments, header declares {arity[name]}"


def shim_ops():
    return set(re.findall(r"PD_BUILD_OP\((\w+)\)", SHIM.read_text())) | {"spmm_sum", "spmm_mean", "spmm_min", "spmm_max"}


def test_every_seam_of_integration_md_has_an_op():
    """INTEGRATION.md section 2 tells a maintainer which paddle_sparse_ops.<op> to call at
    every seam; each of them must be registered by the shim."""
    text = INTEGRATION.read_text()
    seam = text[text.index("| file:line | today | with the HIP ops |"):text.index("`paddle_sparse_amd/` in this repository is that host layer")]
    named = set(re.findall(r"`(?:paddle_sparse_ops\.)?([a-z_0-9]+)(?:\([^`]*\))?`", seam))
    ops = shim_ops()
    wanted = {n for n in named if n in ops or n in {
        "index_sort", "make_keys", "make_keys_checked", "sort_pairs", "split_keys", "bincount", "count2ptr",
        "invert_permutation", "unique_sorted", "segment_csr_perm", "scatter", "ind2ptr", "gather_rows", "merge_sorted",
        "coalesce", "csr_row_stats", "transpose_weights", "spmm_value_bw", "spmm_sum_bw_csc", "spmm_minmax_bw_csc",
        "spmm_minmax_bw", "csc_edge_tags", "sample_adj"}}
    assert len(wanted) >= 20, sorted(wanted)
    assert not (wanted - ops), f"named in INTEGRATION.md but not registered: {sorted(wanted - ops)}"


def test_every_entry_point_is_bound_or_listed_as_unbound():
    """Each C-ABI function is either called by the shim or named in INTEGRATION.md's
    'not bound by the shim' list with the reason."""
    text = INTEGRATION.read_text()
    unbound = text[text.index("**Not bound by the shim**"):]
    unbound = unbound[:unbound.index("\n\n")]
    called = set(re.findall(r"\b(psa_[a-z0-9_]+)\s*\(", _strip_comments(SHIM.read_text())))
    for name in declared_functions():
        assert name in called or f"`{name}`" in unbound, f"{name} is neither bound nor listed as unbound"


def test_shim_type_checks():
    """g++ -fsyntax-only of the shim against include/paddle_sparse_hip.h and the test-only
    stub of the custom-op API (tests/paddle_stub): pointer types, argument order, arity."""
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    res = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", f"-I{ROOT / 'tests' / 'paddle_stub'}",
                          f"-I{ROOT / 'include'}", str(SHIM)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


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


# ---- the documented build command on a clean checkout ----------------------------
def _clean_copy(tmp_path):
    import shutil

    dst = tmp_path / "checkout"
    ignore = shutil.ignore_patterns("__pycache__", "lib", "*.so", "*.o")
    shutil.copytree(ROOT / "paddle_sparse_amd", dst / "paddle_sparse_amd", ignore=ignore)
    shutil.copytree(ROOT / "include", dst / "include")
    assert not (dst / "paddle_sparse_amd" / "lib").exists()
    return dst


def test_documented_build_command_runs_on_a_clean_checkout(tmp_path):
    """`python -m paddle_sparse_amd.build` is what README, INTEGRATION.md and the ImportError
    text tell a user to run.  `-m` imports the package first, and the package loads the
    library it is about to build: the command must work when no library exists yet.  A stub
    `hipcc` (writes its -o file) stands in for the compiler: the test is about the import
    order, and the real compile is __graft_entry__.build()'s job."""
    import os
    import subprocess
    import sys

    dst = _clean_copy(tmp_path)
    bindir = tmp_path / "bin"
    bindir.mkdir()
    stub = bindir / "hipcc"
    stub.write_text('#!/bin/sh\nout=""\nwhile [ $# -gt 0 ]; do\n  if [ "$1" = "-o" ]; then out="$2"; shift; fi\n  shift\ndone\n'
                    '[ -n "$out" ] && : > "$out"\nexit 0\n')
    stub.chmod(0o755)
    env = dict(os.environ, PATH=f"{bindir}:{os.environ['PATH']}", PYTHONPATH="")
    res = subprocess.run([sys.executable, "-m", "paddle_sparse_amd.build"], cwd=dst, env=env, capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lib = dst / "paddle_sparse_amd" / "lib" / "libpaddle_sparse_hip.so"
    assert lib.exists() and res.stdout.strip().endswith("libpaddle_sparse_hip.so")
    objs = sorted(p.name for p in (dst / "build" / "hip").glob("*.o"))
    assert objs == sorted(p.stem + ".o" for p in (ROOT / "paddle_sparse_amd" / "csrc").glob("*.hip"))


def test_importing_the_package_without_the_library_fails_loudly(tmp_path):
    """No library, no product: the import raises and names the build command (there is no
    CPU or eager fallback to fall into) — also for a plain `import`, not only for `-m build`."""
    import os
    import subprocess
    import sys

    dst = _clean_copy(tmp_path)
    env = dict(os.environ, PYTHONPATH="")
    for code in ("import paddle_sparse_amd", "from paddle_sparse_amd import ops", "import paddle_sparse_amd.distributed"):
        res = subprocess.run([sys.executable, "-c", code], cwd=dst, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode != 0
        assert "ImportError" in res.stderr and "python -m paddle_sparse_amd.build" in res.stderr
