"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` runs on a CPU-only box: oracle vs golden vectors, host logic,
C-ABI symbol checks.  `-m gpu` tests are the parity tests proper and call the
HIP kernels through the C-ABI on a real MI355X.
"""
import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP kernels run)")


@pytest.fixture(scope="session")
def kats():
    with open(ROOT / "tests" / "golden" / "reference_kats.json") as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (gcc) and the HIP core (hipcc) are built."""
    import importlib.util

    import oracle

    spec = importlib.util.spec_from_file_location("psa_build", ROOT / "paddle_sparse_amd" / "build.py")
    hip_build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hip_build)
    oracle.build()
    hip_build.build()
    # PSA_SPMM_VARIANT=<n> runs the whole suite with one SpMM kernel variant forced
    # (e.g. 30 = the edge-balanced forward wherever its shapes allow)
    variant = os.environ.get("PSA_SPMM_VARIANT")
    if variant:
        from paddle_sparse_amd import ops

        ops.spmm_set_variant(int(variant))
    yield
