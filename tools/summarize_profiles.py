"""Condense gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>/: the kernel-trace stats CSV,
the PMC traffic of the fused step kernel with the calibration that corrects it, and the bench line."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = Path("gpurun_out") / tag, Path("profiles") / tag
dst.mkdir(parents=True, exist_ok=True)


def counters(d, kname):
    f = glob.glob(str(src / d / "*" / "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if kname in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}


shutil.copy(glob.glob(str(src / "trace" / "*" / "*kernel_stats.csv"))[0], dst / "bench_kernel_stats.csv")
shutil.copy(src / "bench_plain.log", dst / "bench_stdout.log")
bench = json.loads(open(src / "bench_plain.log").read().strip().splitlines()[-1])
step = "poker_step_kernel"
fetch = counters("fetch", step)["FETCH_SIZE"]
write = counters("write", step)["WRITE_SIZE"]
cal_r = counters("calib_fetch", "calib_read_kernel")["FETCH_SIZE"]["mean"]
cal_w = counters("calib_write", "calib_write_kernel")["WRITE_SIZE"]["mean"]
cal_bytes = 512 * 1024 * 1024
read_corr = cal_bytes / (cal_r * 1024)
write_corr = cal_bytes / (cal_w * 1024)
stats = {r["Name"]: r for r in csv.DictReader(open(dst / "bench_kernel_stats.csv"))}
step_row = next(v for k, v in stats.items() if step in k)
out = {
    "tag": tag,
    "kernel": step_row["Name"].split("(")[0],
    "rocprof_kernel_trace": {"calls": int(step_row["Calls"]), "avg_us": float(step_row["AverageNs"]) / 1e3,
                             "min_us": float(step_row["MinNs"]) / 1e3, "max_us": float(step_row["MaxNs"]) / 1e3},
    "bench_event_chunk_avg_us": bench["roofline"]["kernel_us"],
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "calibration": {"stream_bytes": cal_bytes, "FETCH_SIZE_KB": cal_r, "WRITE_SIZE_KB": cal_w,
                    "read_correction": read_corr, "write_correction": write_corr,
                    "how": "tools/pmc_calibrate.py: 512 MiB read / written with one dword per lane, the access shape of the step kernel"},
    "traffic_bytes_per_launch": (fetch["mean"] * read_corr + write["mean"] * write_corr) * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "tables_per_launch": bench["config"]["tables_per_gpu"],
    "sq": counters("sq", step) if (src / "sq").exists() else None,
    "bench": bench,
}
json.dump(out, open(dst / "step_kernel_profile.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("rocprof_kernel_trace", "bench_event_chunk_avg_us", "traffic_bytes_per_launch",
                                      "algorithmic_bytes_per_launch")}, indent=1))

# trainer loop (learner in the loop): kernel-trace stats of the same command + the plain lines
tr = glob.glob(str(src / "trainer" / "*" / "*kernel_stats.csv"))
if tr:
    shutil.copy(tr[0], dst / "trainer_kernel_stats.csv")
    lines = {}
    for name in ("trainer_plain", "trainer_reference_loop", "trainer_torch_learner"):
        f = src / f"{name}.log"
        if f.exists():
            last = [ln for ln in f.read_text().splitlines() if ln.startswith("{")]
            if last:
                lines[name] = json.loads(last[-1])
    json.dump(lines, open(dst / "trainer_lines.json", "w"), indent=1)
    print({k: round(v["value"] / 1e6, 1) for k, v in lines.items()}, "M env-steps/s")
