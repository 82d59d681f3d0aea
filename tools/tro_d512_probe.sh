#!/bin/bash
# Tile height of the fused transposed store where the LDS ring has room for it: C2 cropped to 512 depth bins, 16-row tiles
# (64-byte segments of the D x H image) against 32-row tiles (128-byte segments), row-major and two-pass beside them.
# usage (through gpurun): bash tools/tro_d512_probe.sh   (needs tools/mkvariant.sh tr32 -DFUSED_TR_ROWS=32 -DFUSED_TR_RING=40)
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
out=gpurun_out/tro_d512_probe.txt
: > $out
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" python3 bench.py --steps ${AB_STEPS:-400} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $BARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-34s %.1f M A-scans/s  step %.4f ms  frac %.4f  %s W %s MHz  parity %s' % ('$label', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac'], (d.get('power') or {}).get('package_w_last_half'), (d.get('power') or {}).get('sclk_mhz_avg'), d['parity'].get('worst_db_err_over_tol', d['parity'])))
" | tee -a $out
}
S="$root/fdoct_amd/libfdoct_hip_single.so"; T="$root/fdoct_amd/libfdoct_hip_tr32.so"
for round in 1 2; do
  for dp in 512 256; do
    BARGS="--display-points $dp" run "r$round D$dp rowmajor" FDOCT_LIB="$S"
    BARGS="--display-points $dp --layout transposed" run "r$round D$dp transposed two-pass" FDOCT_LIB="$S" FDOCT_NO_TRO=1
    BARGS="--display-points $dp --layout transposed" run "r$round D$dp transposed fused 16 rows" FDOCT_LIB="$S"
    BARGS="--display-points $dp --layout transposed" run "r$round D$dp transposed fused 32 rows" FDOCT_LIB="$T"
  done
done
