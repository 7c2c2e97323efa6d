#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 passes behind the numbers bench.py and profiles/ quote.
#   tools/collect_profiles.sh <round-tag> [parts]      e.g. r04   (parts: poker envs trainer harness bench rehearsal; default all)
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never mixed
# with --stats / trace domains).  Raw output goes to gpurun_out/<tag>/, summaries to profiles/<tag>/.
# Poker workloads: 65,536 tables (BASELINE config 2, the bench default) and 1,048,576 tables (config 4's total on one GPU),
# each with active_players sampled 2..10 and forced to 10.
set -e
T="timeout -k 10 300"      # a profiled bench takes seconds; never let one hang the box
TAG=${1:-r03}
PARTS=${2:-"poker envs trainer harness bench rehearsal"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --inproc --no-cpu-baseline --trainer-loop off --other-envs off --census off --min-timed-ms 50"
P="rocprofv3 --kernel-trace --output-format csv"
if [[ $PARTS == *poker* ]]; then
# ... and at 65,536 tables once more in the DRIVER's form (--steps 20 --warmup 5: blocks of 20 steps cut some launches to one
# check interval, ~8-9 steps per launch instead of ~10), so that the driver line finds counter traffic of its own launch mix
for F in long:65536 long:1048576 driver:65536; do N=${F#*:}; for A in sampled 10; do
  if [[ $F == driver:* ]]; then S="--tables $N --steps 20 --warmup 5 --min-timed-ms 300 --active-players $A"; K=${N}_${A}_driver
  else S="--tables $N --steps 600 --warmup 100 --active-players $A"; K=${N}_$A; fi
  rm -rf $OUT/trace_$K $OUT/fetch_$K $OUT/write_$K $OUT/sq_$K $OUT/sqw_$K
  echo "[collect] $(date +%T) $K: kernel trace"; $T $P --stats -d $OUT/trace_$K -- $B $S > $OUT/bench_trace_$K.log 2>&1
  echo "[collect] $(date +%T) $K: FETCH_SIZE"; $T $P --pmc FETCH_SIZE -d $OUT/fetch_$K -- $B $S > $OUT/bench_fetch_$K.log 2>&1
  echo "[collect] $(date +%T) $K: WRITE_SIZE"; $T $P --pmc WRITE_SIZE -d $OUT/write_$K -- $B $S > $OUT/bench_write_$K.log 2>&1
  echo "[collect] $(date +%T) $K: SQ"; $T $P --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/sq_$K -- $B $S > $OUT/bench_sq_$K.log 2>&1
  echo "[collect] $(date +%T) $K: SQ waits"; $T $P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/sqw_$K -- $B $S > $OUT/bench_sqw_$K.log 2>&1
  echo "[collect] $(date +%T) $K: plain"; $T $B $S > $OUT/bench_plain_$K.log 2> $OUT/bench_plain_$K.err
done; done
echo "[collect] $(date +%T) calibration"; rm -rf $OUT/calib_fetch $OUT/calib_write
$T $P --pmc FETCH_SIZE -d $OUT/calib_fetch -- python tools/pmc_calibrate.py > $OUT/calib_fetch.log 2>&1
$T $P --pmc WRITE_SIZE -d $OUT/calib_write -- python tools/pmc_calibrate.py > $OUT/calib_write.log 2>&1
fi
if [[ $PARTS == *envs* ]]; then
  rm -rf $OUT/envs_trace $OUT/envs_fetch $OUT/envs_write $OUT/envs_sq
  echo "[collect] $(date +%T) other environments: kernel trace"; $T $P --stats -d $OUT/envs_trace -- python tools/bench_envs.py > $OUT/envs_trace.log 2>&1
  echo "[collect] $(date +%T) other environments: FETCH_SIZE"; $T $P --pmc FETCH_SIZE -d $OUT/envs_fetch -- python tools/bench_envs.py > $OUT/envs_fetch.log 2>&1
  echo "[collect] $(date +%T) other environments: WRITE_SIZE"; $T $P --pmc WRITE_SIZE -d $OUT/envs_write -- python tools/bench_envs.py > $OUT/envs_write.log 2>&1
  echo "[collect] $(date +%T) other environments: SQ"; $T $P --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d $OUT/envs_sq -- python tools/bench_envs.py > $OUT/envs_sq.log 2>&1
  echo "[collect] $(date +%T) other environments: plain"; $T python tools/bench_envs.py > $OUT/envs_lines.jsonl 2> $OUT/envs_lines.err
  $T python tools/qtable_steps.py 262144 130 --fused > $OUT/qtable_steps_fused.jsonl 2>&1
  $T python tools/qtable_steps.py 262144 130 > $OUT/qtable_steps_separate.jsonl 2>&1
fi
if [[ $PARTS == *trainer* ]]; then
  # the bench's own trainer_loop leg (same tool, episodes and warm-up as the number in the bench line) under the kernel trace
  rm -rf $OUT/trainer
  echo "[collect] $(date +%T) trainer loop"; $T $P --stats -d $OUT/trainer -- python bench.py --inproc --no-cpu-baseline --trainer-loop on --trainer-tables-large 0 --other-envs off --census off --active-players sampled --steps 200 --warmup 40 --min-timed-ms 10 > $OUT/trainer_trace.log 2>&1
  $T python bench.py --inproc --no-cpu-baseline --trainer-loop on --other-envs off --census off --active-players sampled --steps 200 --warmup 40 --min-timed-ms 10 > $OUT/trainer_plain.log 2> $OUT/trainer_plain.err
fi
if [[ $PARTS == *trainer* ]]; then
  # ... and the same loop at the reference's published size, with the SQ counters of its kernels (tools/trainer_profile_at.sh)
  echo "[collect] $(date +%T) trainer loop at 2,000,000 tables"
  bash tools/trainer_profile_at.sh 2000000 4 $TAG/trainer_2m counters > $OUT/trainer_2m.log 2>&1 || echo "trainer 2M profile failed"
  mkdir -p profiles/$TAG
  cp $OUT/trainer_2m/kernel_stats.csv profiles/$TAG/trainer_kernel_stats_2000000.csv 2>/dev/null
  cp $OUT/trainer_2m/counters.txt profiles/$TAG/trainer_counters_2000000.txt 2>/dev/null
  grep '"value"' $OUT/trainer_2m/plain.log > profiles/$TAG/trainer_line_2000000.json 2>/dev/null
fi
if [[ $PARTS == *harness* ]]; then
  # the reference's benchmark harness on the HIP classes (benchmarking/Poker/run.py): the four presets' reports
  mkdir -p profiles/$TAG
  for PRE in quick standard stress mi355x; do
    echo "[collect] $(date +%T) harness preset $PRE"
    rm -rf $OUT/harness_$PRE
    $T python -m pulselib_amd.benchmarking --preset $PRE --output-dir $OUT/harness_$PRE > $OUT/harness_$PRE.txt 2> $OUT/harness_$PRE.err || echo "harness $PRE failed"
    grep -v amdgpu.ids $OUT/harness_$PRE.txt > profiles/$TAG/harness_$PRE.txt
    f=$(ls $OUT/harness_$PRE/*.json 2>/dev/null | head -1); [ -n "$f" ] && cp $f profiles/$TAG/harness_$PRE.json
  done
fi
if [[ $PARTS == *bench* ]]; then
  echo "[collect] $(date +%T) default bench (the driver's line)"; $T python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
  $T python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.log 2> $OUT/bench_driver_style.err
fi
if [[ $PARTS == *rehearsal* ]]; then
  # several ranks on ONE device (gloo): what a one-GPU box can rehearse of --gpus N
  for R in "2 host" "2 shm" "3 shm"; do set -- $R
    echo "[collect] $(date +%T) rehearsal: $1 ranks, $2 exchange"
    PULSE_BENCH_ONE_DEVICE=1 $T python bench.py --gpus $1 --steps 600 --warmup 100 --min-timed-ms 100 --stop-exchange $2 > $OUT/rehearsal_${1}ranks_$2.log 2> $OUT/rehearsal_${1}ranks_$2.err || echo "rehearsal $R failed"
  done
fi
python tools/summarize_profiles.py $TAG
