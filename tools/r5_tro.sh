#!/bin/bash
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "f64_frames or transposed or tiles_that" > gpurun_out/r5_tests_tro.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_tests_tro.log
tail -6 gpurun_out/r5_tests_tro.log
AB_ROUNDS=2 AB_ARGS="--layout transposed" bash tools/ab.sh base base@FDOCT_TRO_RING=20 base@FDOCT_PRECISE_DIVISION=0 base@FDOCT_PRECISE_DIVISION=0,FDOCT_TRO_RING=20 > gpurun_out/r5_ab_tro.log 2>&1
AB_ROUNDS=1 AB_ARGS="--layout transposed --background-2d" bash tools/ab.sh base base@FDOCT_TRO_RING=20 >> gpurun_out/r5_ab_tro.log 2>&1
AB_ROUNDS=1 AB_ARGS="--layout transposed --display-points 512" bash tools/ab.sh base base@FDOCT_TRO_RING=40 >> gpurun_out/r5_ab_tro.log 2>&1
AB_ROUNDS=1 bash tools/ab.sh base >> gpurun_out/r5_ab_tro.log 2>&1
cat gpurun_out/r5_ab_tro.log
