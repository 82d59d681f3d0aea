#!/bin/bash
# A/B of the known-zero input blocks of the zero-pad stage's inverse transform (FDOCT_WAVE_ZPRUNE): the five shipped shapes,
# built-in and run-time compiled kernels, two interleaved rounds
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r5_zprune.txt
: > $out
for round in 1 2; do
  echo "== round $round: ZPRUNE=0" >> $out
  FDOCT_LIB=$PWD/fdoct_amd/libfdoct_hip_z0.so FDOCT_JIT_DEFINES="-DFDOCT_WAVE_ZPRUNE=0" FDOCT_JIT_CACHE=/tmp/jz0 python3 tools/bench_generic.py 1.0 2>&1 | grep "A-scans/s\|built-in" >> $out
  echo "== round $round: ZPRUNE=1" >> $out
  FDOCT_JIT_CACHE=/tmp/jz1 python3 tools/bench_generic.py 1.0 2>&1 | grep "A-scans/s\|built-in" >> $out
done
cut -c1-120 $out
