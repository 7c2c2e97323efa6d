"""Timeline of a kernel trace (rocprofv3 --kernel-trace --output-format csv): per kernel the count / median duration, and
how the wall time between the first and the last launch splits into kernels and idle gaps.
    python tools/trace_gaps.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 4        # drop warm-up launches
rows = rows[skip:]
d = collections.defaultdict(list)
busy, gaps, last_end = 0, [], rows[0][0]
for s, e, n in rows:
    d[n.split("(")[0].replace("void (anonymous namespace)::", "")[:60]].append((e - s) / 1e3)
    if s > last_end:
        gaps.append((s - last_end) / 1e3)
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
wall = (rows[-1][1] - rows[0][0]) / 1e3
print(f"{len(rows)} launches over {wall / 1e3:.2f} ms: busy {busy / 1e3 / wall * 100:.1f} %, idle {100 - busy / 1e3 / wall * 100:.1f} % in {len(gaps)} gaps "
      f"(median {sorted(gaps)[len(gaps) // 2]:.1f} us, mean {sum(gaps) / len(gaps):.1f} us)")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"  {k:60s} n={len(v):5d} total {sum(v) / 1e3:8.2f} ms  median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}")
