#!/bin/bash
# Where the fused transposed store's time goes: the full kernel against measurement builds that skip the global stores of the
# write-out (-DFDOCT_TRO_X=1: ring, counters and LDS reads stay) or the whole write-out step (-DFDOCT_TRO_X=2: ring and
# counters stay), for the distributed write-out and the last-arriver one (-DFDOCT_TRO_DW=2), at 1024 and 256 depth bins.
# (The measurement builds produce no image: parity is not checked for them.)
# usage (through gpurun): bash tools/tro_cost_probe.sh
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
out=gpurun_out/tro_cost_probe.txt
: > $out
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" python3 bench.py --steps ${AB_STEPS:-400} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $BARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %.1f M A-scans/s  step %.4f ms  %s W %s MHz  = %.4f M per MHz' % ('$label', d['value']/1e6, d['roofline']['kernel_ms_avg'], (d.get('power') or {}).get('package_w_last_half'), (d.get('power') or {}).get('sclk_mhz_avg'), d['value']/1e6/((d.get('power') or {}).get('sclk_mhz_avg') or 1)))
" | tee -a $out
}
L="$root/fdoct_amd/libfdoct_hip_"
for dp in 1024 256; do
  BARGS="--display-points $dp" run "D$dp rowmajor" FDOCT_LIB="${L}single.so"
  for v in single x1 x2 la lax1 lax2; do
    BARGS="--display-points $dp --layout transposed" run "D$dp fused $v" FDOCT_LIB="${L}$v.so"
  done
done
