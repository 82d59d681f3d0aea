#!/bin/bash
# C1 (1024 samples -> numfftpoints 1024, the 512-point plan: four rows per wave) in the reference's D x H layout: the chain's own
# transposed store against the row-major chain and the two-pass route, under wave counts, ring sizes and both division settings.
# usage: gpurun -- 'bash tools/c1_tro_probe.sh [base|words|groups]'   -> gpurun_out/r6_c1_tro_probe.txt
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
run() { label="$1"; shift; python3 bench.py --workload C1 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 --steps 300 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %.1f M A-scans/s  %.4f ms  %s W' % ('$label', d['value']/1e6, d['ms_per_step'], (d.get('power') or {}).get('package_w_last_half')))
"; }
case ${1:-base} in
  base)
    run "rowmajor 8 waves"
    run "rowmajor 6 waves" --threads-per-block 384
    run "rowmajor 5 waves" --threads-per-block 320
    run "transposed (default)" --layout transposed
    run "transposed one word" --layout transposed --one-word-division
    FDOCT_TRO_RING=20 run "transposed ring 20" --layout transposed
    run "transposed 4 waves" --layout transposed --threads-per-block 256
    FDOCT_NO_TRO=1 run "transposed two-pass" --layout transposed ;;
  words)
    run "transposed one word 6 waves" --layout transposed --one-word-division
    run "transposed one word 5 waves" --layout transposed --one-word-division --threads-per-block 320
    run "transposed one word 4 waves" --layout transposed --one-word-division --threads-per-block 256
    run "transposed both words (default)" --layout transposed
    run "transposed both words 4 waves" --layout transposed --threads-per-block 256
    run "transposed both words 3 waves" --layout transposed --threads-per-block 192 ;;
  groups)
    # groups of eight waves (32-row tiles, 128-byte segments: a library built with -DFDOCT_TRO_GROUP_WAVES=8 as fdoct_amd/libfdoct_hip_gw8.so)
    # against the shipped groups of four, interleaved -> profiles/r06_c1_group_ab.txt
    G4=$PWD/fdoct_amd/libfdoct_hip.so; G8=$PWD/fdoct_amd/libfdoct_hip_gw8.so
    for round in 1 2; do
      FDOCT_LIB=$G4 run "r$round rowmajor"
      FDOCT_LIB=$G4 run "r$round transposed, groups of 4 waves (16-row tiles)" --layout transposed
      FDOCT_LIB=$G8 run "r$round transposed, groups of 8 waves (32-row tiles)" --layout transposed
      FDOCT_LIB=$G4 run "r$round transposed one word, groups of 4" --layout transposed --one-word-division
      FDOCT_LIB=$G8 run "r$round transposed one word, groups of 8" --layout transposed --one-word-division
    done ;;
esac
