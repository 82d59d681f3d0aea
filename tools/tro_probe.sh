#!/bin/bash
# The transposed (D x H) output written by the chain itself (fused_kernel TRO) under bench.py, as a matrix of
# (label, library variant, environment, bench arguments) runs -- ONE script for the probes that used to be six.
# usage (through gpurun):  bash tools/tro_probe.sh [preset] [variant ...]      -> gpurun_out/tro_<preset>_probe.txt
#   base [variants]  row-major, the two-pass path (FDOCT_NO_TRO=1) and the fused store of each library variant   (default)
#   align    H = 1000 against H = 1024 (segments aligned to their own size), 16- and 32-row tiles      (variants single, tr32)
#   cost     measurement builds without the write-out's stores / steps, 1024 and 256 bins   (single x1 x2 la lax1 lax2)
#   d512     16- against 32-row tiles where the ring has room, 512 / 256 bins                         (single, tr32)
#   final    distributed write-out against the last-arriver form, 1024 / 512 / 256 bins              (single, la)
#   slack    rings of 20 / 28 / 40 slots at 512 bins                                                  (single rs28 rs40 la lars40)
# Variants are libfdoct_hip_<name>.so built with tools/mkvariant.sh (flags in DESIGN.md 3.1a / profiles/r03_tro_*.txt);
# "base" is the shipped library.
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
mkdir -p gpurun_out
preset=${1:-base}; [ $# -gt 0 ] && shift
out=gpurun_out/tro_${preset}_probe.txt
: > "$out"
lib() { [ "$1" = base ] && echo "$root/fdoct_amd/libfdoct_hip.so" || echo "$root/fdoct_amd/libfdoct_hip_$1.so"; }
run() {  # label, variant, "ENV=.. ENV2=.." (or -), bench args...
  local label=$1 v=$2 envs=$3; shift 3
  [ "$envs" = - ] && envs=""
  # shellcheck disable=SC2086
  env FDOCT_LIB="$(lib "$v")" $envs python3 bench.py --steps ${AB_STEPS:-400} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d.get('power') or {}
        print('%-46s %.1f M A-scans/s  step %.4f ms  frac %.4f  %s W %s MHz  parity %s' % ('$label', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac'], p.get('package_w_last_half'), p.get('sclk_mhz_avg'), d['parity'].get('worst_db_err_over_tol', d['parity'])))
" | tee -a "$out"
}
T="--layout transposed"
for round in 1 2; do
  case $preset in
    base)
      run "r$round rowmajor base" base -
      run "r$round transposed two-pass" base FDOCT_NO_TRO=1 $T
      for v in "${@:-base}"; do run "r$round transposed fused $v" "$v" - $T; done ;;
    align)
      for h in 1000 1024; do
        run "r$round H$h D1024 rowmajor" single - --lines-per-frame $h
        run "r$round H$h D1024 fused 16 rows" single - --lines-per-frame $h $T
        run "r$round H$h D512 fused 16 rows" single - --lines-per-frame $h --display-points 512 $T
        run "r$round H$h D512 fused 32 rows" tr32 - --lines-per-frame $h --display-points 512 $T
      done ;;
    cost)
      [ $round = 2 ] && break
      for dp in 1024 256; do
        run "D$dp rowmajor" single - --display-points $dp
        for v in single x1 x2 la lax1 lax2; do run "D$dp fused $v" $v - --display-points $dp $T; done
      done ;;
    d512)
      for dp in 512 256; do
        run "r$round D$dp rowmajor" single - --display-points $dp
        run "r$round D$dp transposed two-pass" single FDOCT_NO_TRO=1 --display-points $dp $T
        run "r$round D$dp transposed fused 16 rows" single - --display-points $dp $T
        run "r$round D$dp transposed fused 32 rows" tr32 - --display-points $dp $T
      done ;;
    final)
      for dp in 1024 512 256; do
        run "r$round D$dp rowmajor" single - --display-points $dp
        run "r$round D$dp two-pass" single FDOCT_NO_TRO=1 --display-points $dp $T
        run "r$round D$dp fused, distributed write-out (shipped)" single - --display-points $dp $T
        run "r$round D$dp fused, last arriver (-DFDOCT_TRO_DW=2)" la - --display-points $dp $T
      done ;;
    slack)
      run "r$round D512 rowmajor" single - --display-points 512
      for v in single rs28 rs40 la lars40; do run "r$round D512 fused $v" $v - --display-points 512 $T; done ;;
    *) echo "unknown preset $preset"; exit 1 ;;
  esac
done
