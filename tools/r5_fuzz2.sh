#!/bin/bash
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r5_fuzz2_summary.txt
: > $out
run() { echo "== $*" >> $out; timeout -k 10 560 python3 tools/fuzz_parity.py "$@" > gpurun_out/r5_fuzz2.log 2>&1; echo "exit $?" >> $out; grep -E "^FAIL|^noise|failures:" gpurun_out/r5_fuzz2.log | tail -8 >> $out; }
run 5301 400 0.5 0 0 0.3 0 0 0
run 5302 150 0 0.6 0 0.2 0 0 0
run 5303 400 0 0 0.5 0.5 0.3 0.3 0.3
cat $out
