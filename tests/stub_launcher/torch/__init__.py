"""Test stub (tests/test_bench_launcher.py): stands in for torch in the LAUNCHER process of bench.py and for
`python -m torch.distributed.run` in its child, so the launcher's argv, exit-code and stdout handling can be checked
on a machine with no GPU.  Not product code."""
import os


class cuda:  # noqa: N801
    @staticmethod
    def device_count() -> int:
        return int(os.environ.get("STUB_GPUS", "0"))

    @staticmethod
    def is_available():
        raise AssertionError("the launcher process must not initialise the GPU")
