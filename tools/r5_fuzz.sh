#!/bin/bash
# Round 5's seeded sweeps; ONE summary file per call (profiles/r05_fuzz_summary.txt collects them).
# usage (gpurun --timeout 1200): bash tools/r5_fuzz.sh [set]      set = first (default; SEEDS / COUNT override), other, window, jit
# Each run line = tools/fuzz_parity.py <seed> <cases> <shares: jit, long rows, forced routes, weak frames, tall frames, device pointers, reuse>
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
set=${1:-first}
out=gpurun_out/r5_fuzz_${set}_summary.txt
: > $out
run() {
  echo "== $*" >> $out
  timeout -k 10 ${LIMIT:-540} python3 tools/fuzz_parity.py "$@" > gpurun_out/r5_fuzz_$1.log 2>&1
  echo "exit $?" >> $out
  grep -E "^FAIL|^noise|^window|failures:" gpurun_out/r5_fuzz_$1.log | tail -12 >> $out
}
case $set in
  first)   # every dimension at once (run twice in the round: seeds 5101-5103 and 5201-5203)
    for seed in ${SEEDS:-5101 5102 5103}; do LIMIT=1000 run $seed ${COUNT:-300} 0.2 0.1 0.2 0.3 0.2 0.2 0.2; done ;;
  other)   # other mixes: half run-time compiled; mostly long rows; half forced routes and weak frames
    run 5301 400 0.5 0 0 0.3 0 0 0
    run 5302 150 0 0.6 0 0.2 0 0 0
    run 5303 400 0 0 0.5 0.5 0.3 0.3 0.3 ;;
  window)  # after the "window" class (D << N / 2) went into tests/fuzz_cases.py: 5303 again and one more seed
    run 5303 400 0 0 0.5 0.5 0.3 0.3 0.3
    run 5304 300 0.3 0 0.5 0.3 0.5 0.3 0.3 ;;
  jit)     # after the wave-per-row kernels got their depth bound and zero-block rules: run-time compiled geometries in most cases
    run 5401 400 0.7 0 0.2 0.3 0.2 0.2 0.2
    run 5402 300 0.2 0.1 0.3 0.3 0.2 0.2 0.2 ;;
  *) echo "unknown set $set"; exit 2 ;;
esac
cat $out
