#!/bin/bash
# SQ counters of the learner kernels in the bench's trainer-loop leg at 65,536 tables (their own pass: no trace domains).
#   tools/trainer_counters.sh <out-dir-under-gpurun_out>
OUT=gpurun_out/${1:-trainer_counters}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/a -- python bench.py --inproc --no-cpu-baseline --trainer-loop on --trainer-tables-large 0 --other-envs off --census off --active-players sampled --steps 200 --warmup 40 --min-timed-ms 10 > $OUT.a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/b -- python bench.py --inproc --no-cpu-baseline --trainer-loop on --trainer-tables-large 0 --other-envs off --census off --active-players sampled --steps 200 --warmup 40 --min-timed-ms 10 > $OUT.b.log 2>&1 || echo "second pass failed"
python - $OUT <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    fs = glob.glob(f"{sys.argv[1]}/{sub}/*/*counter_collection.csv")
    if not fs:
        print(sub, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "qnet" not in k:
            continue
        k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        w = d.get("SQ_WAVES", 0) or 1
        print(k, {c: round(v / w, 1) for c, v in d.items()}, "(per wavefront; SQ_WAVES summed over launches:", int(w), ")")
PY
