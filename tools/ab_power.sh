#!/bin/bash
# Like ab.sh, one round, but prints the package power / clock / A-scans per joule of each variant (bench.py's "power" object).
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
for v in "$@"; do
  lib="$root/fdoct_amd/libfdoct_hip_$v.so"; [ "$v" = base ] && lib="$root/fdoct_amd/libfdoct_hip.so"
  FDOCT_LIB="$lib" python3 bench.py --steps ${AB_STEPS:-2000} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['power']
        print('%-10s %.1f M A-scans/s  %.0f GB/s  %.0f W  %.0f MHz  %.3g uJ per A-scan (package)' % ('$v', d['value']/1e6, d['roofline']['achieved'], p['package_w_last_half'], p['sclk_mhz_avg'], 1e6*p['package_w_last_half']/d['value']))
"
done
