"""GPU: the drop-in boundary used the way the Paddle shim would use it — a
C++ program with only the HIP runtime and libpaddle_sparse_hip.so in the
process (no torch, no Python): ind2ptr, spmm and the coalesce chain on the
reference's known answers, and the error path."""
import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def test_cpp_client_of_the_c_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib_dir = ROOT / "paddle_sparse_amd" / "lib"
    exe = tmp_path / "cabi_client"
    build = subprocess.run(
        [hipcc, "-O1", "-std=c++17", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cabi_client.cpp"),
         f"-L{lib_dir}", "-lpaddle_sparse_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)],
        capture_output=True, text=True)
    assert build.returncode == 0, build.stdout + build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, f"exit {run.returncode}\n{run.stdout}\n{run.stderr}"
    assert "OK" in run.stdout
