#!/bin/bash
# A/B of library variants on one GPU box, interleaved rounds (rule: compare within one process/box).
# usage (through gpurun): bash tools/ab.sh variantA variantB ...   (fdoct_amd/libfdoct_hip_<variant>.so; "base" = the shipped library)
# The variant is selected with FDOCT_LIB (fdoct_amd/capi.py::library_path): the shipped .so is never overwritten.
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
rounds=${AB_ROUNDS:-3}
for round in $(seq 1 "$rounds"); do
  for v in "$@"; do
    lib="$root/fdoct_amd/libfdoct_hip_$v.so"; [ "$v" = base ] && lib="$root/fdoct_amd/libfdoct_hip.so"
    [ -f "$lib" ] || { echo "missing $lib"; exit 1; }
    FDOCT_LIB="$lib" python3 bench.py --steps ${AB_STEPS:-600} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $round %-10s %.1f M A-scans/s  kernel %.4f ms  frac %.4f  parity %s' % ('$v', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['parity'].get('worst_db_err_over_tol', d['parity'])))
"
  done
done
