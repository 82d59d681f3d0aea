#!/bin/bash
# fourth sweep of round 5: after the wave-per-row kernels got their depth bound and zero-block rules (run-time compiled geometries
# drawn in most cases: every width / multiplier / length / depth is another set of pruned blocks) and the staged-mode fix
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r5_fuzz4_summary.txt
: > $out
run() { echo "== $*" >> $out; timeout -k 10 540 python3 tools/fuzz_parity.py "$@" > gpurun_out/r5_fuzz4.log 2>&1; echo "exit $?" >> $out; grep -E "^FAIL|^noise|^window|failures:" gpurun_out/r5_fuzz4.log | tail -8 >> $out; }
run 5401 400 0.7 0 0.2 0.3 0.2 0.2 0.2
run 5402 300 0.2 0.1 0.3 0.3 0.2 0.2 0.2
cat $out
