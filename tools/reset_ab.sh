#!/bin/bash
# A/B of the reset kernel's store forms on ONE box: the product library against the diagnostic twin
# built by `make -C pulselib_amd/csrc reset-narrow` (the stores of rounds 1-3), at the sizes where the kernel is memory-bound.  On the GPU box:
#   tools/reset_ab.sh <out-file-under-gpurun_out>
OUT=gpurun_out/${1:-reset_ab.txt}
for N in 65536 131072 1048576; do
  for rnd in 0 1; do
    for L in libpulse_hip_reset_narrow.so libpulse_hip.so; do
      echo "== $L N=$N round $rnd" >> $OUT
      PULSE_LIB=$PWD/pulselib_amd/$L timeout -k 10 120 python3 tools/time_reset.py $N cache-on-only >> $OUT 2>&1 || echo "FAILED" >> $OUT
    done
  done
done
cat $OUT
