"""Test stub for `python -m torch.distributed.run`: records its argv and the launcher-relevant environment, prints what
a real N-rank run prints on stdout (a banner line that is not the result, then rank 0's JSON line) and exits with
STUB_RC."""
import json
import os
import sys
import time

if __name__ == "__main__":
    with open(os.environ["STUB_RECORD"], "w") as f:
        json.dump({"argv": sys.argv[1:], "world_size_in_env": "WORLD_SIZE" in os.environ}, f)
    time.sleep(float(os.environ.get("STUB_SLEEP", "0")))
    print("RCCL version banner (not the result line)")
    for _ in range(int(os.environ.get("STUB_LINES", "1"))):
        print(json.dumps({"metric": "spmm_sum_gedges_per_s", "value": 1.0, "n_gpus": int(os.environ.get("STUB_N", "2"))}))
    print("stub stderr chatter", file=sys.stderr)
    sys.exit(int(os.environ.get("STUB_RC", "0")))
