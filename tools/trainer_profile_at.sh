#!/bin/bash
# Kernel trace + SQ counters of the trainer loop (tools/bench_trainer.py = bench.py's trainer_loop leg: same environment,
# learner, opponents and episode loop) at a given table count.  Run on the GPU box:
#   tools/trainer_profile_at.sh <tables> <episodes> <out-dir-under-gpurun_out> [counters]
N=${1:-2000000}; E=${2:-4}; OUT=gpurun_out/${3:-trainer_at_$N}; WITH_PMC=${4:-counters}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT; mkdir -p $OUT
CMD="python3 tools/bench_trainer.py --tables $N --episodes $E --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --stats -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:72]:72s} {r["Calls"]:>7s} {float(r["AverageNs"]) / 1000:9.2f} us {r["Percentage"]:>6s} %')
PY
grep '"value"' $OUT/trace.log
timeout -k 10 300 $CMD > $OUT/plain.log 2>&1 && grep '"value"' $OUT/plain.log
[ "$WITH_PMC" = counters ] || exit 0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { echo "counter pass a failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || echo "counter pass b failed"
python3 - $OUT <<'PY' | tee $OUT/counters.txt
import csv, glob, sys, collections
for sub in ("a", "b"):
    fs = glob.glob(f"{sys.argv[1]}/{sub}/*/*counter_collection.csv")
    if not fs:
        print(sub, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); calls[k] += 1
    for k, d in acc.items():
        w = d.get("SQ_WAVES", 0) or 1
        print(k[:80], "launches", calls[k], {c: round(v / w, 1) for c, v in d.items()}, "(per wavefront; SQ_WAVES per launch:", round(w / max(calls[k], 1), 1), ")")
PY
