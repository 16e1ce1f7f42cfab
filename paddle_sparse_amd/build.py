"""Build the C-ABI HIP core (libpaddle_sparse_hip.so) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  Objects go to build/ (git-ignored), the
shared library to paddle_sparse_amd/lib/ (git-ignored, but it travels with the
tree).  Run:  python -m paddle_sparse_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "paddle_sparse_amd" / "csrc"
INCLUDE = ROOT / "include"
OBJ_DIR = ROOT / "build" / "hip"
LIB_DIR = ROOT / "paddle_sparse_amd" / "lib"
LIB_PATH = LIB_DIR / "libpaddle_sparse_hip.so"

ARCH = "gfx950"
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-fPIC",
    "-std=c++17",
    "-Wall",
    "-Wno-unused-function",
    "-fno-gpu-rdc",
    "-munsafe-fp-atomics",  # native global_atomic_add_f32 (no CAS loop)
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP core cannot be built")
    return exe


def _newest_header_mtime() -> float:
    hdrs = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    return max(p.stat().st_mtime for p in hdrs)


def _compile(src: Path, force: bool, verbose: bool) -> Path:
    obj = OBJ_DIR / (src.stem + ".o")
    dep_mtime = max(src.stat().st_mtime, _newest_header_mtime())
    if not force and obj.exists() and obj.stat().st_mtime >= dep_mtime:
        return obj
    extra = os.environ.get("PSA_EXTRA_HIPCC_FLAGS", "").split()  # A/B builds of a kernel variant (-D...)
    cmd = [hipcc(), *HIPCC_FLAGS, *extra, f"-I{INCLUDE}", f"-I{CSRC}", "-c", str(src),
           "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, jobs: int = 6) -> Path:
    """Compile every csrc/*.hip for gfx950 and link the shared library."""
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError(f"no HIP sources under {CSRC}")
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, verbose), srcs))
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest_obj:
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC",
               "-o", str(LIB_PATH), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(path)
