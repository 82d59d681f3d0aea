#!/bin/bash
# The shipped transposed store (distributed write-out, ring of 20 slots at 1024 depth bins and 40 up to 512) against the
# write-out by the wave that completes a tile (-DFDOCT_TRO_DW=2), the two-pass path and row-major, at 1024 / 512 / 256 depth bins.
# usage (through gpurun): bash tools/tro_final_probe.sh   (needs tools/mkvariant.sh single; tools/mkvariant.sh la -DFDOCT_TRO_DW=2)
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
out=gpurun_out/tro_final_probe.txt
: > $out
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" python3 bench.py --steps ${AB_STEPS:-400} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $BARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-44s %.1f M A-scans/s  step %.4f ms  %s W %s MHz  parity %s' % ('$label', d['value']/1e6, d['roofline']['kernel_ms_avg'], (d.get('power') or {}).get('package_w_last_half'), (d.get('power') or {}).get('sclk_mhz_avg'), d['parity'].get('worst_db_err_over_tol', d['parity'])))
" | tee -a $out
}
L="$root/fdoct_amd/libfdoct_hip_"
for round in 1 2; do
  for dp in 1024 512 256; do
    BARGS="--display-points $dp" run "r$round D$dp rowmajor" FDOCT_LIB="${L}single.so"
    BARGS="--display-points $dp --layout transposed" run "r$round D$dp two-pass" FDOCT_LIB="${L}single.so" FDOCT_NO_TRO=1
    BARGS="--display-points $dp --layout transposed" run "r$round D$dp fused, distributed write-out (shipped)" FDOCT_LIB="${L}single.so"
    BARGS="--display-points $dp --layout transposed" run "r$round D$dp fused, last arriver (-DFDOCT_TRO_DW=2)" FDOCT_LIB="${L}la.so"
  done
done
