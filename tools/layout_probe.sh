#!/bin/bash
# Transposed (D x H, the reference's bscan layout) output: chain + transpose pass over chunks of B-scans whose row-major
# intermediate is FDOCT_TR_CHUNK_MB large (0 = the whole batch in one chunk), with nt and with plain intermediate stores.
# usage (through gpurun): bash tools/layout_probe.sh [variant ...]   -> gpurun_out/layout_probe.txt
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
out=gpurun_out/layout_probe.txt
: > $out
vars="${@:-base}"
for round in 1 2; do
for v in $vars; do
  lib="$root/fdoct_amd/libfdoct_hip_$v.so"; [ "$v" = base ] && lib="$root/fdoct_amd/libfdoct_hip.so"
  for mb in 0 16 32 64 128 256; do
    FDOCT_LIB="$lib" FDOCT_TR_CHUNK_MB=$mb python3 bench.py --layout transposed --steps 400 --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $round %-8s chunk %4s MB  %.1f M A-scans/s  step %.4f ms  frac %.4f  parity %s' % ('$v', '$mb', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['parity'].get('worst_db_err_over_tol', d['parity'])))
" | tee -a $out
  done
done
done
