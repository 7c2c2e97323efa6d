"""Timeline of a kernel trace (rocprofv3 --kernel-trace --output-format csv): the launches are split into segments at
pauses > 300 us (bench phases); per segment with > 200 launches: GPU-busy share, time per kernel, and the idle gaps by
the kernel that follows them.
    python tools/trace_gaps.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    base = n.split("(")[0]
    if "poker_step_kernel" in base:
        return "poker_step_kernel" + base[base.index("<"):][:40]
    return base[:50]


segs, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-5:]) > 300_000:
        segs.append(cur)
        cur = [r]
    else:
        cur.append(r)
segs.append(cur)
for s in [s for s in segs if len(s) > 200]:
    wall = s[-1][1] - s[0][0]
    busy, last, gaps = 0, s[0][0], []
    per = collections.defaultdict(lambda: [0, 0.0])
    for a, b, n in s:
        if a > last:
            gaps.append(((a - last) / 1e3, short(n)))
        busy += max(0, b - max(a, last))
        last = max(last, b)
        per[short(n)][0] += 1
        per[short(n)][1] += (b - a) / 1e3
    print(f"segment of {len(s)} launches, wall {wall / 1e3:.0f} us, GPU busy {busy / wall * 100:.1f} %")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:60s} n={v[0]:5d} total {v[1]:9.1f} us ({v[1] * 1e3 / wall * 100:5.1f} %) avg {v[1] / v[0]:.1f}")
    by = collections.defaultdict(lambda: [0, 0.0])
    for g, n in gaps:
        by[n][0] += 1
        by[n][1] += g
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"   gap before {k:49s} n={v[0]:5d} total {v[1]:9.1f} us ({v[1] * 1e3 / wall * 100:5.1f} %) avg {v[1] / v[0]:.1f}")
