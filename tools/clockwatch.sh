#!/bin/bash
# Samples rocm-smi clocks/power while the bench runs (through gpurun).
cd "$GRAFT_REPO_ROOT"
rocm-smi --showmaxpower --showpower --showclocks > gpurun_out/smi_idle.txt 2>&1
python3 bench.py --steps 20000 --warmup 5 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 "$@" > gpurun_out/clockwatch_bench.log 2>&1 &
bp=$!
sleep 6
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|mclk|fclk|Power|power" ; sleep 2; done > gpurun_out/smi_load.txt
wait $bp
tail -1 gpurun_out/clockwatch_bench.log | cut -c1-160
