"""Condense gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>/: the kernel-trace stats CSVs, the PMC
traffic of the chunk kernel with the calibration that corrects it, the SQ counters, and the bench lines."""
import collections
import csv
import glob
import os
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = Path("gpurun_out") / tag, Path("profiles") / tag
dst.mkdir(parents=True, exist_ok=True)
STEP = "poker_step_kernel"


def counters(d, kname):
    f = sorted(glob.glob(str(src / d / "*" / "*counter_collection.csv")), key=os.path.getmtime, reverse=True)   # newest run first
    if not f:
        return {}
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if kname in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}


def last_json(path):
    lines = [ln for ln in Path(path).read_text().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


cal_r = counters("calib_fetch", "calib_read_kernel").get("FETCH_SIZE", {}).get("mean")
cal_w = counters("calib_write", "calib_write_kernel").get("WRITE_SIZE", {}).get("mean")
cal_bytes = 512 * 1024 * 1024
calibration = None
if cal_r and cal_w:
    calibration = {"stream_bytes": cal_bytes, "FETCH_SIZE_KB": cal_r, "WRITE_SIZE_KB": cal_w, "read_correction": cal_bytes / (cal_r * 1024),
                   "write_correction": cal_bytes / (cal_w * 1024),
                   "how": "tools/pmc_calibrate.py: 512 MiB read / written with one dword per lane (the access shape of most of the kernel's loads and stores)"}

for n in (65536, 1048576):
    stats = sorted(glob.glob(str(src / f"trace_{n}" / "*" / "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    if not stats:
        continue
    shutil.copy(stats[0], dst / f"bench_kernel_stats_{n}.csv")
    rows = {r["Name"]: r for r in csv.DictReader(open(stats[0]))}
    step_row = next(v for k, v in rows.items() if STEP in k)
    bench = last_json(src / f"bench_plain_{n}.log")
    fetch = counters(f"fetch_{n}", STEP).get("FETCH_SIZE")
    write = counters(f"write_{n}", STEP).get("WRITE_SIZE")
    out = {
        "tag": tag, "tables_per_launch": n, "steps_per_launch": 5,
        "kernel": step_row["Name"].split("(")[0],
        "rocprof_kernel_trace": {"calls": int(step_row["Calls"]), "avg_us": float(step_row["AverageNs"]) / 1e3,
                                 "min_us": float(step_row["MinNs"]) / 1e3, "max_us": float(step_row["MaxNs"]) / 1e3},
        "bench_event_avg_us": bench["roofline"]["kernel_us"] if bench else None,
        "algorithmic_bytes_per_launch": 453 * n * 5,
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "calibration": calibration,
        "sq": {**counters(f"sq_{n}", STEP), **counters(f"sqw_{n}", STEP)},
        "bench": bench,
    }
    if fetch and write and calibration:
        out["traffic_bytes_per_launch"] = (fetch["mean"] * calibration["read_correction"] + write["mean"] * calibration["write_correction"]) * 1024
        out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    out["frac_of_8TBs_by_rocprof_avg"] = out["algorithmic_bytes_per_launch"] / (out["rocprof_kernel_trace"]["avg_us"] * 1e-6) / 8e12
    name = "step_kernel_profile.json" if n == 65536 else f"step_kernel_profile_{n}.json"
    json.dump(out, open(dst / name, "w"), indent=1)
    print(n, json.dumps({k: out.get(k) for k in ("rocprof_kernel_trace", "bench_event_avg_us", "traffic_bytes_per_launch", "algorithmic_bytes_per_launch",
                                                  "traffic_over_algorithmic", "frac_of_8TBs_by_rocprof_avg")}, indent=1))

for name in ("bench_default", "bench_driver_style"):
    f = src / f"{name}.log"
    if f.exists():
        shutil.copy(f, dst / f"{name}.json")
tr = sorted(glob.glob(str(src / "trainer" / "*" / "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if tr:
    shutil.copy(tr[0], dst / "trainer_kernel_stats.csv")
    line = last_json(src / "trainer_plain.log")
    if line:
        json.dump(line, open(dst / "trainer_line.json", "w"), indent=1)
