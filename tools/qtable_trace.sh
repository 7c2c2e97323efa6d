#!/bin/bash
# Per-launch kernel durations of the fused Q-learning roll-out step along a roll-out from reset (tools/qtable_steps.py under
# the kernel trace), optionally with timing-only ablations.   tools/qtable_trace.sh <out-dir-under-gpurun_out> [ablate flags ...]
OUT=gpurun_out/${1:-qtable_trace}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
for AB in 0 "$@"; do
  rm -rf $OUT/t$AB
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$AB -- python3 tools/qtable_steps.py 262144 60 --fused --ablate=$AB > $OUT/t$AB.log 2>&1 || { echo "ablate $AB failed"; tail -3 $OUT/t$AB.log; continue; }
  python3 - $OUT/t$AB $AB <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000) for r in rows if "qtable" in r["Kernel_Name"]]
out = []
for i, (k, d) in enumerate(seq):
    if "rollout_step" in k:
        nd = seq[i + 1] if i + 1 < len(seq) else None
        out.append((round(d, 1), round(nd[1], 1) if nd and "defer" in nd[0] else None))
pick = [0, 1, 2, 3, 4, 5, 8, 12, 16, 20, 30, 40, 59]
print("ablate", sys.argv[2], "(rollout us, follow-up us) at steps", pick, ":", [out[i] for i in pick if i < len(out)])
w = out[6:36]
print("   steps 6..35 mean: rollout %.1f + follow-up %.1f us" % (sum(a for a, b in w) / len(w), sum(b or 0 for a, b in w) / len(w)))
PY
done
