#!/bin/bash
# Ring slack of the fused transposed store: 512 depth bins, 16-row tiles, rings of 20 / 28 / 40 slots (4 / 12 / 24 rows may be
# deposited while a tile is written out), distributed and last-arriver write-out.
# usage (through gpurun): bash tools/tro_slack_probe.sh
#   needs tools/mkvariant.sh rs28 -DFUSED_TR_RING=28; rs40 -DFUSED_TR_RING=40; lars40 -DFUSED_TR_RING=40 -DFDOCT_TRO_DW=2; la -DFDOCT_TRO_DW=2
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
out=gpurun_out/tro_slack_probe.txt
: > $out
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" python3 bench.py --steps ${AB_STEPS:-400} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $BARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %.1f M A-scans/s  step %.4f ms  %s W %s MHz  parity %s' % ('$label', d['value']/1e6, d['roofline']['kernel_ms_avg'], (d.get('power') or {}).get('package_w_last_half'), (d.get('power') or {}).get('sclk_mhz_avg'), d['parity'].get('worst_db_err_over_tol', d['parity'])))
" | tee -a $out
}
L="$root/fdoct_amd/libfdoct_hip_"
for round in 1 2; do
  BARGS="--display-points 512" run "r$round D512 rowmajor" FDOCT_LIB="${L}single.so"
  for v in single rs28 rs40 la lars40; do
    BARGS="--display-points 512 --layout transposed" run "r$round D512 fused $v" FDOCT_LIB="${L}$v.so"
  done
done
