"""HIP branch for the reference's setup_ops.py (setup_ops.py:29-38,70-125).

NOT runnable in this repository's image (Paddle is absent).  It shows the
build a paddle_sparse maintainer adds so that `paddle_sparse_ops` becomes the
shim in this directory linked against libpaddle_sparse_hip.so:

    FORCE_HIP=1 PSA_ROOT=/path/to/this/repo python setup_ops_hip.py install

The reference detects GPUs with paddle.device.cuda.device_count() + CUDA_HOME
and picks nvcc arch flags from paddle.version.cuda_version (setup_ops.py:29-32,
54-67); neither applies on ROCm, so the HIP branch keys on FORCE_HIP /
paddle.is_compiled_with_rocm() and needs no arch list: the kernels are
prebuilt for gfx950 inside libpaddle_sparse_hip.so (python -m
paddle_sparse_amd.build) and the shim is plain host C++.
"""
import os
import os.path as osp

import paddle
from paddle.utils.cpp_extension import CppExtension, setup

PSA_ROOT = os.environ.get("PSA_ROOT", osp.abspath(osp.join(osp.dirname(__file__), "..", "..")))
WITH_HIP = os.getenv("FORCE_HIP", "0") == "1" or paddle.is_compiled_with_rocm()
assert WITH_HIP, "this branch builds the MI355X shim only"

lib_dir = osp.join(PSA_ROOT, "paddle_sparse_amd", "lib")

setup(
    name="paddle_sparse_ops",
    ext_modules=CppExtension(
        sources=[osp.join(osp.dirname(__file__), "paddle_sparse_hip_ops.cc")],
        include_dirs=[osp.join(PSA_ROOT, "include")],
        library_dirs=[lib_dir],
        libraries=["paddle_sparse_hip"],
        runtime_library_dirs=[lib_dir],
        define_macros=[("WITH_PYTHON", None), ("WITH_HIP", None)],
        extra_compile_args={"cxx": ["-O3", "-Wno-sign-compare"]},
    ),
)
