#!/bin/bash
# Kernel trace of the bench's trainer-loop leg at 65,536 tables; prints the per-kernel averages.  Run on the GPU box:
#   tools/trainer_trace.sh <out-dir-under-gpurun_out>
OUT=gpurun_out/${1:-trainer_trace}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --stats -d $OUT -- python bench.py --inproc --no-cpu-baseline --trainer-loop on --trainer-tables-large 0 --other-envs off --census off --active-players sampled --steps 200 --warmup 40 --min-timed-ms 10 > $OUT.log 2>&1 || exit 1
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
python - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:64]:64s} {r["Calls"]:>7s} {float(r["AverageNs"]) / 1000:9.2f} us {r["Percentage"]:>6s} %')
PY
grep "trainer loop" $OUT.log
