#!/bin/bash
# the other BASELINE workloads, both division settings
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
: > gpurun_out/r5_wl.log
for wl in C4 C3 C1; do
  AB_ROUNDS=1 AB_STEPS=300 AB_ARGS="--workload $wl" bash tools/ab.sh base base@FDOCT_PRECISE_DIVISION=0 2>&1 | sed "s/^/$wl /" >> gpurun_out/r5_wl.log
done
cat gpurun_out/r5_wl.log
