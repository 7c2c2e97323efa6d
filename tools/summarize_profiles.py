"""Condense gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>/: the kernel-trace stats CSVs, the PMC
traffic of the chunk kernel with the calibration that corrects it, the SQ counters, and the bench lines.  Every Poker
summary records the workload (tables per launch, steps per launch, active_players mode) and the sha256 of the kernel
sources it was collected from: bench.py quotes `traffic` only from a summary whose workload AND sources match its build."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src, dst = Path("gpurun_out") / tag, Path("profiles") / tag
dst.mkdir(parents=True, exist_ok=True)
STEP = "poker_step_kernel"
sys.path.insert(0, ".")
import bench  # noqa: E402  (KERNEL_SOURCES, the same hash bench.py checks)

sha = hashlib.sha256()
for rel in bench.KERNEL_SOURCES:
    sha.update(Path(rel).read_bytes())
SOURCE_SHA = sha.hexdigest()[:16]


def newest(pattern):
    f = sorted(glob.glob(str(pattern)), key=os.path.getmtime, reverse=True)
    return f[0] if f else None


def counters(d, kname):
    f = newest(src / d / "*" / "*counter_collection.csv")
    if not f:
        return {}
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if kname in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}


def last_json(path):
    if not Path(path).exists():
        return None
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

for n, form in ((65536, ""), (1048576, ""), (65536, "_driver")):
    for mode in ("sampled", "10"):
        k = f"{n}_{mode}{form}"
        stats = newest(src / f"trace_{k}" / "*" / "*kernel_stats.csv")
        if not stats:
            continue
        shutil.copy(stats, dst / f"bench_kernel_stats_{k}.csv")
        rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
        step_rows = [v for kk, v in rows.items() if STEP in kk]
        step_row = max(step_rows, key=lambda r: float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["Calls"]))
        bench_line = last_json(src / f"bench_plain_{k}.log")
        spl = bench_line["roofline"]["steps_per_launch"] if bench_line and bench_line.get("roofline") else 10
        fetch = counters(f"fetch_{k}", STEP).get("FETCH_SIZE")
        write = counters(f"write_{k}", STEP).get("WRITE_SIZE")
        alg = 453 * n * spl
        out = {
            "tag": tag, "tables_per_launch": n, "steps_per_launch": int(round(spl)), "steps_per_launch_mean": spl, "active_players": mode,
            "kernel_source_sha256_16": SOURCE_SHA, "kernel_sources": list(bench.KERNEL_SOURCES),
            "kernel": step_row["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0],
            "rocprof_kernel_trace": {"calls": int(step_row["Calls"]), "avg_us": float(step_row["AverageNs"]) / 1e3,
                                     "min_us": float(step_row["MinNs"]) / 1e3, "max_us": float(step_row["MaxNs"]) / 1e3},
            "other_kernels_avg_us": {kk.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][-48:]: float(v["AverageNs"]) / 1e3
                                     for kk, v in rows.items() if STEP not in kk and "poker" in kk},
            "bench_event_avg_us": bench_line["roofline"]["kernel_us"] if bench_line and bench_line.get("roofline") else None,
            "algorithmic_bytes_per_launch": alg,
            "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "calibration": calibration,
            "sq": {**counters(f"sq_{k}", STEP), **counters(f"sqw_{k}", STEP)},
            "bench": bench_line,
        }
        if fetch and write and calibration:
            out["traffic_bytes_per_launch"] = (fetch["mean"] * calibration["read_correction"] + write["mean"] * calibration["write_correction"]) * 1024
            out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / alg
        out["frac_of_8TBs_by_rocprof_avg"] = alg / (out["rocprof_kernel_trace"]["avg_us"] * 1e-6) / 8e12
        sq = out["sq"]
        if "SQ_INSTS_VALU" in sq and "SQ_WAVES" in sq and sq["SQ_WAVES"]["mean"]:
            waves = sq["SQ_WAVES"]["mean"]
            out["per_wavefront_step"] = {c: sq[c]["mean"] / waves / spl for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if c in sq}
        json.dump(out, open(dst / f"step_kernel_profile_{k}.json", "w"), indent=1)
        print(k, json.dumps({kk: out.get(kk) for kk in ("rocprof_kernel_trace", "bench_event_avg_us", "traffic_bytes_per_launch", "algorithmic_bytes_per_launch",
                                                        "traffic_over_algorithmic", "frac_of_8TBs_by_rocprof_avg", "per_wavefront_step", "other_kernels_avg_us")}, indent=1))

for name in ("bench_default", "bench_driver_style"):
    f = src / f"{name}.log"
    if f.exists():
        shutil.copy(f, dst / f"{name}.json")
tr = newest(src / "trainer" / "*" / "*kernel_stats.csv")
if tr:
    shutil.copy(tr, dst / "trainer_kernel_stats.csv")
    line = last_json(src / "trainer_plain.log")
    if line:
        json.dump({"trainer_loop": line.get("trainer_loop"), "value_env_only": line.get("value")}, open(dst / "trainer_line.json", "w"), indent=1)
ev = newest(src / "envs_trace" / "*" / "*kernel_stats.csv")
if ev:
    shutil.copy(ev, dst / "envs_kernel_stats.csv")
    summary = {}
    for kn in ("qtable_rollout_step_kernel", "qtable_select_kernel", "qtable_update_kernel", "qtable_defer_kernel", "tfe_step4_kernel", "tfe_step_kernel", "particle2d_step_kernel",
               "blackjack_step_kernel"):
        c = {}
        for d, names in (("envs_fetch", ("FETCH_SIZE",)), ("envs_write", ("WRITE_SIZE",)),
                         ("envs_sq", ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY"))):
            got = counters(d, kn)
            for nm in names:
                if nm in got:
                    c[nm] = got[nm]
        if c:
            if calibration and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                c["traffic_bytes_per_launch"] = (c["FETCH_SIZE"]["mean"] * calibration["read_correction"] + c["WRITE_SIZE"]["mean"] * calibration["write_correction"]) * 1024
            summary[kn] = c
    json.dump(summary, open(dst / "envs_counters.json", "w"), indent=1)
for name in ("envs_lines.jsonl", "qtable_steps_fused.jsonl", "qtable_steps_separate.jsonl"):
    if (src / name).exists():
        shutil.copy(src / name, dst / name)
for f in glob.glob(str(src / "rehearsal_*")):
    shutil.copy(f, dst / os.path.basename(f))
print("summaries in", dst)
