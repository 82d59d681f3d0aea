#!/bin/bash
# third sweep of round 5, re-run after the "window" class (D << N / 2) was added to tests/fuzz_cases.py, + one more seed
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r5_fuzz3_summary.txt
: > $out
run() { echo "== $*" >> $out; timeout -k 10 540 python3 tools/fuzz_parity.py "$@" > gpurun_out/r5_fuzz3.log 2>&1; echo "exit $?" >> $out; grep -E "^FAIL|^noise|^window|failures:" gpurun_out/r5_fuzz3.log | tail -8 >> $out; }
run 5303 400 0 0 0.5 0.5 0.3 0.3 0.3
run 5304 300 0.3 0 0.5 0.3 0.5 0.3 0.3
cat $out
