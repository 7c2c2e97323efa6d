#!/bin/bash
# Kernel-trace averages + SQ counters of the 2048 step kernel at 262,144 boards (tools/bench_envs.py).  On the GPU box:
#   tools/tfe_kernel_time.sh <out-dir-under-gpurun_out>
OUT=gpurun_out/${1:-tfe_time}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --stats -d $OUT/trace -- python3 tools/bench_envs.py > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("tfe", "qtable", "particle", "blackjack_step")):
        print(f'{r["Name"].replace("(anonymous namespace)::", "")[:70]:70s} {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1000:8.2f} min {float(r["MinNs"]) / 1000:8.2f} max {float(r["MaxNs"]) / 1000:8.2f} us')
PY
