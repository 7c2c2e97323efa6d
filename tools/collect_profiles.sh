#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 passes behind the numbers bench.py and profiles/ quote.
#   tools/collect_profiles.sh <round-tag>      e.g. r01
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never mixed
# with --stats / trace domains).  Raw output goes to gpurun_out/<tag>/, summaries to profiles/<tag>/.
set -e
T="timeout -k 10 300"      # a profiled bench takes seconds; never let one hang the box
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "[collect] $(date +%T) next pass"; $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --inproc --no-cpu-baseline > $OUT/bench_trace.log 2>&1
echo "[collect] $(date +%T) next pass"; $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --inproc --no-cpu-baseline --steps 600 --warmup 100 > $OUT/bench_fetch.log 2>&1
echo "[collect] $(date +%T) next pass"; $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --inproc --no-cpu-baseline --steps 600 --warmup 100 > $OUT/bench_write.log 2>&1
echo "[collect] $(date +%T) next pass"; $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python tools/pmc_calibrate.py > $OUT/calib_fetch.log 2>&1
echo "[collect] $(date +%T) next pass"; $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python tools/pmc_calibrate.py > $OUT/calib_write.log 2>&1
echo "[collect] $(date +%T) next pass"; $T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python bench.py --inproc --no-cpu-baseline --steps 600 --warmup 100 > $OUT/bench_sq.log 2>&1
echo "[collect] $(date +%T) trainer loop"; $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trainer -- python tools/bench_trainer.py --episodes 20 > $OUT/trainer_trace.log 2>&1
echo "[collect] trainer loop, plain"; $T python tools/bench_trainer.py --episodes 20 > $OUT/trainer_plain.log 2>&1
$T python tools/bench_trainer.py --episodes 5 --loop reference > $OUT/trainer_reference_loop.log 2>&1
$T python tools/bench_trainer.py --episodes 10 --learner torch > $OUT/trainer_torch_learner.log 2>&1
echo "[collect] plain bench"; $T python bench.py > $OUT/bench_plain.log 2>&1
python tools/summarize_profiles.py $TAG
