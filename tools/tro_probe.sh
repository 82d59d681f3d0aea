#!/bin/bash
# Transposed (D x H) output written by the chain itself (fused_kernel TRO) against the two-pass path (FDOCT_NO_TRO=1) and the
# row-major headline, library variants side by side (tile rows, write-out steps in flight).
# usage (through gpurun): bash tools/tro_probe.sh [variant ...]   -> gpurun_out/tro_probe.txt
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
out=gpurun_out/tro_probe.txt
: > $out
vars="${@:-base}"
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" python3 bench.py --steps ${AB_STEPS:-600} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $BARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-28s %.1f M A-scans/s  step %.4f ms  frac %.4f  %s W %s MHz  parity %s' % ('$label', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac'], (d.get('power') or {}).get('package_w_last_half'), (d.get('power') or {}).get('sclk_mhz_avg'), d['parity'].get('worst_db_err_over_tol', d['parity'])))
" | tee -a $out
}
for round in 1 2; do
  BARGS="" run "r$round rowmajor base" FDOCT_LIB="$root/fdoct_amd/libfdoct_hip.so"
  BARGS="--layout transposed" run "r$round transposed two-pass" FDOCT_LIB="$root/fdoct_amd/libfdoct_hip.so" FDOCT_NO_TRO=1
  for v in $vars; do
    lib="$root/fdoct_amd/libfdoct_hip_$v.so"; [ "$v" = base ] && lib="$root/fdoct_amd/libfdoct_hip.so"
    BARGS="--layout transposed" run "r$round transposed fused $v" FDOCT_LIB="$lib"
  done
done
