"""Summarize `rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES -- python tools/ablate_prof.py`:
instructions per wavefront of every phase-ablated instantiation of the step kernel (the mask is in the kernel name)."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"poker_step_kernel<(\d+)u, true, 4, 3, false, false>", r["Kernel_Name"])
    if m:
        acc[int(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {0x1FF: "full", 0x1FD: "-equity", 0x1DF: "-showdown", 0x1DD: "-equity-showdown", 0x17F: "-reward", 0x0FF: "-obs", 0x15D: "-eq-sd-reward",
         0x05D: "-eq-sd-reward-obs", 0x001: "capture only"}
full = None
for mask in sorted(acc, reverse=True):
    d = acc[mask]
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    row = {c: sum(v) / len(v) / w for c, v in d.items() if c != "SQ_WAVES"}
    if mask == 0x1FF:
        full = row
    print(f"{names.get(mask, hex(mask)):22s} " + "  ".join(f"{c} {x:8.1f}" for c, x in sorted(row.items())) +
          ("" if full is None else f"   VALU vs full {row['SQ_INSTS_VALU'] - full['SQ_INSTS_VALU']:+7.1f}"))
