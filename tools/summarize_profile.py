#!/usr/bin/env python3
"""Turn gpurun_out/prof_<round>/ (rocprofv3 CSVs) into the small, committed
summaries under profiles/:
  <round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats (names trimmed)
  <round>_pmc.json           FETCH_SIZE / WRITE_SIZE / TCC hit-miss per launch
  traffic.json               per-launch HBM-side bytes that bench.py reports
Calibration (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are KiB; on
gfx950 FETCH_SIZE under-reports wide reads, so the read side is scaled by
known_bytes / reported_bytes measured on the permutation-matrix SpMM (every B
row read exactly once, B > Infinity Cache) in the same access pattern.
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / f"prof_{rnd}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= 110 else name[:107] + "..."


# kernel stats
stats = sorted(glob.glob(str(src / "stats" / "*" / "*_kernel_stats.csv")), key=lambda f: Path(f).stat().st_mtime)
if stats:
    rows = list(csv.DictReader(open(stats[-1])))  # newest run
    with open(dst / f"{rnd}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])


def pmc(dirname: str, match: str):
    out = collections.defaultdict(list)
    files = sorted(glob.glob(str(src / dirname / "*" / "*_counter_collection.csv")),
                   key=lambda f: Path(f).stat().st_mtime)
    for f in files[-1:]:  # newest run only
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


M = N = 2_000_000
F = 128
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "spmm_fused_kernel"
res = {"units": "FETCH_SIZE/WRITE_SIZE in KiB per launch (mean over launches)", "kernel_match": KERNEL}
cal_f, _ = pmc("pmc_fetch_calib", KERNEL)
cal_w, _ = pmc("pmc_write_calib", KERNEL)
c3_f, n = pmc("pmc_fetch_c3", KERNEL)
c3_w, _ = pmc("pmc_write_c3", KERNEL)
c3_l2, _ = pmc("pmc_l2_c3", KERNEL)
known_read = N * 4 * F + M * 12 + (M + 1) * 8
known_write = M * 4 * F
res["calibration"] = {
    "workload": "permutation-matrix SpMM, M=N=2M, nnz=2M, F=128 (each B row read once; B = 1.024 GB > 256 MiB Infinity Cache)",
    "known_read_bytes": known_read, "known_write_bytes": known_write,
    "FETCH_SIZE_KiB": cal_f.get("FETCH_SIZE"), "WRITE_SIZE_KiB": cal_w.get("WRITE_SIZE"),
}
rf = known_read / (cal_f["FETCH_SIZE"] * 1024)
wf = known_write / (cal_w["WRITE_SIZE"] * 1024)
res["calibration"]["read_factor"] = rf
res["calibration"]["write_factor"] = wf
read_b = c3_f["FETCH_SIZE"] * 1024 * rf
write_b = c3_w["WRITE_SIZE"] * 1024 * wf
res["spmm_sum_c3"] = {
    "launches": n.get("FETCH_SIZE"),
    "FETCH_SIZE_KiB": c3_f["FETCH_SIZE"], "WRITE_SIZE_KiB": c3_w["WRITE_SIZE"],
    "read_bytes_calibrated": read_b, "write_bytes_calibrated": write_b,
    "read_bytes_guide_x2": c3_f["FETCH_SIZE"] * 1024 * 2,
    "hbm_bytes_per_launch": read_b + write_b,
    "TCC_HIT_sum": c3_l2.get("TCC_HIT_sum"), "TCC_MISS_sum": c3_l2.get("TCC_MISS_sum"),
    "l2_hit_rate": c3_l2["TCC_HIT_sum"] / (c3_l2["TCC_HIT_sum"] + c3_l2["TCC_MISS_sum"]),
    "algorithmic_bytes": 20_000_000 * (8 + 4 + 4 * F) + M * (8 + 4 * F),
}
(dst / f"{rnd}_pmc.json").write_text(json.dumps(res, indent=2) + "\n")
traffic = {}
tf = dst / "traffic.json"
if tf.exists():
    traffic = json.loads(tf.read_text())
traffic["spmm_sum_c3"] = {"hbm_bytes_per_launch": round(read_b + write_b), "round": rnd,
                          "source": f"profiles/{rnd}_pmc.json"}
tf.write_text(json.dumps(traffic, indent=2) + "\n")
print(json.dumps(res, indent=2))
