#!/bin/bash
# C1 in D x H: groups of eight waves (32-row tiles, 128-byte segments) against groups of four (16-row tiles, 64-byte segments), interleaved
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
run() { label="$1"; lib="$2"; shift 2; FDOCT_LIB=$lib python3 bench.py --workload C1 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 --steps 300 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-58s %.1f M A-scans/s  %.4f ms  %s W  parity %s' % ('$label', d['value']/1e6, d['ms_per_step'], (d.get('power') or {}).get('package_w_last_half'), d['parity'].get('worst_db_err_over_tol')))
"; }
G8=$PWD/fdoct_amd/libfdoct_hip.so; G4=$PWD/fdoct_amd/libfdoct_hip_gw4.so
for round in 1 2; do
  run "r$round rowmajor" $G8
  run "r$round transposed, groups of 4 waves (16-row tiles)" $G4 --layout transposed
  run "r$round transposed, groups of 8 waves (32-row tiles)" $G8 --layout transposed
  run "r$round transposed one word, groups of 4" $G4 --layout transposed --one-word-division
  run "r$round transposed one word, groups of 8" $G8 --layout transposed --one-word-division
done
