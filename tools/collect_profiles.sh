#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 passes behind the numbers bench.py and profiles/ quote.
#   tools/collect_profiles.sh <round-tag>      e.g. r02
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never mixed
# with --stats / trace domains).  Raw output goes to gpurun_out/<tag>/, summaries to profiles/<tag>/.
# Two workloads: 65,536 tables (BASELINE config 2, the bench default) and 1,048,576 tables (config 4's total on one GPU).
set -e
T="timeout -k 10 300"      # a profiled bench takes seconds; never let one hang the box
TAG=${1:-r02}
OUT=gpurun_out/$TAG
rm -rf $OUT          # a second run into the same directory leaves two sets of CSVs per pass behind
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --inproc --no-cpu-baseline --trainer-loop off"
for N in 65536 1048576; do
  S="--tables $N --steps 600 --warmup 100"
  echo "[collect] $(date +%T) $N tables: kernel trace"; $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$N -- $B $S > $OUT/bench_trace_$N.log 2>&1
  echo "[collect] $(date +%T) $N tables: FETCH_SIZE"; $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$N -- $B $S > $OUT/bench_fetch_$N.log 2>&1
  echo "[collect] $(date +%T) $N tables: WRITE_SIZE"; $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$N -- $B $S > $OUT/bench_write_$N.log 2>&1
  echo "[collect] $(date +%T) $N tables: SQ"; $T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq_$N -- $B $S > $OUT/bench_sq_$N.log 2>&1
  echo "[collect] $(date +%T) $N tables: SQ waits"; $T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sqw_$N -- $B $S > $OUT/bench_sqw_$N.log 2>&1
  echo "[collect] $(date +%T) $N tables: plain"; $T $B $S > $OUT/bench_plain_$N.log 2> $OUT/bench_plain_$N.err
done
echo "[collect] $(date +%T) calibration"; $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python tools/pmc_calibrate.py > $OUT/calib_fetch.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python tools/pmc_calibrate.py > $OUT/calib_write.log 2>&1
echo "[collect] $(date +%T) default bench (the driver's line)"; $T python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
$T python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.log 2> $OUT/bench_driver_style.err
echo "[collect] $(date +%T) trainer loop"; $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trainer -- python tools/bench_trainer.py --episodes 20 > $OUT/trainer_trace.log 2>&1
$T python tools/bench_trainer.py --episodes 20 > $OUT/trainer_plain.log 2>&1
python tools/summarize_profiles.py $TAG
