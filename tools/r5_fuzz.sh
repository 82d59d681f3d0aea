#!/bin/bash
# round 5 sweep: every dimension at once, a few seeds; ONE summary file per run (profiles/r05_fuzz_summary.txt is written once)
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r5_fuzz_summary.txt
: > $out
for seed in ${SEEDS:-5101 5102 5103}; do
  echo "== seed $seed: ${COUNT:-300} cases, shares jit 0.2 long 0.1 forced routes 0.2 weak 0.3 tall 0.2 device pointers 0.2 reuse 0.2" >> $out
  timeout -k 10 1000 python3 tools/fuzz_parity.py $seed ${COUNT:-300} 0.2 0.1 0.2 0.3 0.2 0.2 0.2 > gpurun_out/r5_fuzz_$seed.log 2>&1
  echo "exit $?" >> $out
  grep -E "^FAIL|^noise|failures:" gpurun_out/r5_fuzz_$seed.log | tail -12 >> $out
done
cat $out
