#!/bin/bash
# Does the alignment of the D x H image's rows matter to the fused transposed store?  H = 1000 (rows of the image 4000 B apart:
# segments start at every multiple of 32 B) against H = 1024 (segments aligned to their own size), 16- and 32-row tiles.
# usage (through gpurun): bash tools/tro_align_probe.sh   (needs tools/mkvariant.sh tr32 -DFUSED_TR_ROWS=32 -DFUSED_TR_RING=40)
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
out=gpurun_out/tro_align_probe.txt
: > $out
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" python3 bench.py --steps ${AB_STEPS:-400} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $BARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %.1f M A-scans/s  step %.4f ms  frac %.4f  %s W %s MHz  parity %s' % ('$label', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac'], (d.get('power') or {}).get('package_w_last_half'), (d.get('power') or {}).get('sclk_mhz_avg'), d['parity'].get('worst_db_err_over_tol', d['parity'])))
" | tee -a $out
}
S="$root/fdoct_amd/libfdoct_hip_single.so"; T="$root/fdoct_amd/libfdoct_hip_tr32.so"
for round in 1 2; do
  for h in 1000 1024; do
    BARGS="--lines-per-frame $h" run "r$round H$h D1024 rowmajor" FDOCT_LIB="$S"
    BARGS="--lines-per-frame $h --layout transposed" run "r$round H$h D1024 fused 16 rows" FDOCT_LIB="$S"
    BARGS="--lines-per-frame $h --display-points 512 --layout transposed" run "r$round H$h D512 fused 16 rows" FDOCT_LIB="$S"
    BARGS="--lines-per-frame $h --display-points 512 --layout transposed" run "r$round H$h D512 fused 32 rows" FDOCT_LIB="$T"
  done
done
