#!/bin/bash
# round 5: the whole GPU suite on the two-word default, then A/B of the second word's forms
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_tests.log
tail -8 gpurun_out/r5_tests.log
AB_ROUNDS=3 bash tools/ab.sh base base@FDOCT_PRECISE_DIVISION=0 p32 > gpurun_out/r5_ab.log 2>&1
AB_ROUNDS=2 AB_ARGS=--background-2d bash tools/ab.sh base ibt0 base@FDOCT_PRECISE_DIVISION=0 >> gpurun_out/r5_ab.log 2>&1
AB_ROUNDS=2 AB_ARGS="--layout transposed" bash tools/ab.sh base base@FDOCT_PRECISE_DIVISION=0 >> gpurun_out/r5_ab.log 2>&1
AB_ROUNDS=1 AB_ARGS="--layout transposed --background-2d" bash tools/ab.sh base base@FDOCT_PRECISE_DIVISION=0 >> gpurun_out/r5_ab.log 2>&1
cat gpurun_out/r5_ab.log
