#!/bin/bash
# Where a fused-kernel row's cycles go, phase by phase: measurement builds of ONE instantiation each
# (-DFDOCT_FUSED_PROBE; build lines below) under bench.py; the library prints the shares every 50 launches.
#   PLAN_FLAGS=" " tools/mkvariant.sh probe5 -DFDOCT_DEV_ONE=5 -DFDOCT_FUSED_PROBE                               (C2)
#   PLAN_FLAGS=" " tools/mkvariant.sh probe7 -DFDOCT_DEV_ONE=7 -DFDOCT_DEV_ONE_CPLX=true -DFDOCT_FUSED_PROBE      (C3)
#   PLAN_FLAGS=" " tools/mkvariant.sh probe8 -DFDOCT_DEV_ONE=8 -DFDOCT_DEV_ONE_AVG=1 -DFDOCT_FUSED_PROBE          (C4)
# usage (through gpurun): bash tools/fused_probe.sh [suffix]   -> gpurun_out/fused_probe<suffix>.txt
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
sfx=${1:-}
out=gpurun_out/fused_probe$sfx.txt
: > "$out"
for pair in "probe5$sfx C2" "probe7$sfx C3" "probe8$sfx C4"; do
  set -- $pair
  lib="$root/fdoct_amd/libfdoct_hip_$1.so"
  [ -f "$lib" ] || { echo "missing $lib" | tee -a "$out"; continue; }
  for div in "" "--one-word-division"; do
    echo "== $2 ($1) ${div:-both words (default)}" >> "$out"
    FDOCT_LIB="$lib" python3 bench.py --workload $2 --steps 300 --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 $div 2> gpurun_out/fp.err | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   rate of this measurement build: %.1f M A-scans/s (step %.4f ms)' % (d['value']/1e6, d['roofline']['kernel_ms_avg']))
" >> "$out"
    grep "fused probe" gpurun_out/fp.err | tail -1 | tr '|' '\n' | sed 's/^ */   /' >> "$out"
  done
done
cat "$out"
