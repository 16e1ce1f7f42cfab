"""`python bench.py --gpus N` typed without a launcher starts its own `torch.distributed.run` child (bench.py
self_launch).  Checked here on CPU with a stub `torch` package on PYTHONPATH (tests/stub_launcher): the argv the child
gets, exit-code propagation, that stdout carries exactly rank 0's JSON line, that too few GPUs stops before spawning, and
that the launcher process never imports paddle_sparse_amd (it must stay free of GPU state)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
STUB = ROOT / "tests" / "stub_launcher"

# runs bench.py as __main__ and reports on stderr which modules the LAUNCHER process ended up with
RUNNER = """
import runpy, sys
sys.argv = ["bench.py"] + sys.argv[1:]
rc = 0
try:
    runpy.run_path({bench!r}, run_name="__main__")
except SystemExit as e:
    rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
print("LAUNCHER_MODULES paddle_sparse_amd=%d numpy=%d" % ("paddle_sparse_amd" in sys.modules, "numpy" in sys.modules), file=sys.stderr)
sys.exit(rc)
"""


def run_parent(tmp_path, flags, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(STUB)
    env["STUB_RECORD"] = str(tmp_path / "argv.json")
    env.update({k: str(v) for k, v in env_extra.items()})
    p = subprocess.run([sys.executable, "-c", RUNNER.format(bench=str(ROOT / "bench.py")), *flags],
                       capture_output=True, text=True, env=env, timeout=120)
    rec = json.loads((tmp_path / "argv.json").read_text()) if (tmp_path / "argv.json").exists() else None
    return p, rec


def test_launches_one_child_with_the_same_flags_and_relays_one_line(tmp_path):
    p, rec = run_parent(tmp_path, ["--gpus", "2", "--steps", "7", "--warmup", "3"], STUB_GPUS=8, STUB_N=2)
    assert p.returncode == 0, p.stderr
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 2  # the banner went to stderr
    assert "RCCL version banner" in p.stderr
    a = rec["argv"]
    assert a[:3] == ["--nnodes=1", "--nproc-per-node", "2"]
    assert a[3:5] == ["--master-addr", "127.0.0.1"] and a[5] == "--master-port" and 0 < int(a[6]) < 65536
    assert Path(a[7]) == ROOT / "bench.py"
    assert a[8:] == ["--gpus", "2", "--steps", "7", "--warmup", "3"]
    assert rec["world_size_in_env"] is False  # torch.distributed.run sets it for the ranks, not the launcher
    assert "LAUNCHER_MODULES paddle_sparse_amd=0 numpy=0" in p.stderr


@pytest.mark.parametrize("rc", [1, 17])
def test_child_failure_is_the_parents_exit_code_and_no_line(tmp_path, rc):
    p, _ = run_parent(tmp_path, ["--gpus", "4"], STUB_GPUS=8, STUB_RC=rc)
    assert p.returncode == rc
    assert p.stdout.strip() == ""


def test_too_few_gpus_stops_before_spawning(tmp_path):
    p, rec = run_parent(tmp_path, ["--gpus", "8"], STUB_GPUS=1)
    assert p.returncode != 0 and rec is None
    assert "needs 8 visible GPU(s), this node shows 1" in p.stderr
    assert p.stdout.strip() == ""


def test_rehearsal_on_one_gpu_needs_only_one(tmp_path):
    p, rec = run_parent(tmp_path, ["--gpus", "2"], STUB_GPUS=1, PSA_BENCH_REHEARSE_ON_ONE_GPU=1)
    assert p.returncode == 0 and rec["argv"][2] == "2"


def test_two_result_lines_is_an_error(tmp_path):
    p, _ = run_parent(tmp_path, ["--gpus", "2"], STUB_GPUS=2, STUB_LINES=2)
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_a_child_past_the_limit_is_stopped_and_reported(tmp_path):
    p, _ = run_parent(tmp_path, ["--gpus", "2"], STUB_GPUS=2, STUB_SLEEP=30, PSA_BENCH_LAUNCH_TIMEOUT=1)
    assert p.returncode == 124 and "ran past" in p.stderr and p.stdout.strip() == ""


def test_a_stray_world_size_of_one_is_not_a_launcher(tmp_path):
    """WORLD_SIZE=1 exported by some environment, no LOCAL_RANK: still typed without a launcher."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=str(STUB), STUB_RECORD=str(tmp_path / "argv.json"), STUB_GPUS="4", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, "-c", RUNNER.format(bench=str(ROOT / "bench.py")), "--gpus", "4"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr
    assert json.loads((tmp_path / "argv.json").read_text())["argv"][2] == "4"
