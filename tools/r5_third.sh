#!/bin/bash
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "f64_frames or sim_variant_with_averages or long_rows_and_any or errors_are_loud or weak_fringes_on or transposed_store" > gpurun_out/r5_tests3.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_tests3.log
tail -25 gpurun_out/r5_tests3.log
AB_ROUNDS=2 AB_ARGS="--layout transposed --background-2d" bash tools/ab.sh base base@FDOCT_PRECISE_DIVISION=0 > gpurun_out/r5_ab3.log 2>&1
AB_ROUNDS=1 AB_ARGS="--background-2d" bash tools/ab.sh base base@FDOCT_PRECISE_DIVISION=0 >> gpurun_out/r5_ab3.log 2>&1
cat gpurun_out/r5_ab3.log
