#!/bin/bash
# A/B of library variants on one GPU box, interleaved rounds (rule: compare within one process/box).
# usage (through gpurun): bash tools/ab.sh variantA variantB ...   (fdoct_amd/libfdoct_hip_<variant>.so; "base" = the shipped library)
# A variant may carry environment settings for its runs: name@VAR=VAL[,VAR2=VAL2]  (e.g. base@FDOCT_PRECISE_DIVISION=0)
# The variant is selected with FDOCT_LIB (fdoct_amd/capi.py::library_path): the shipped .so is never overwritten.
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
rounds=${AB_ROUNDS:-3}
for round in $(seq 1 "$rounds"); do
  for spec in "$@"; do
    v=${spec%%@*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*@}
    lib="$root/fdoct_amd/libfdoct_hip_$v.so"; [ "$v" = base ] && lib="$root/fdoct_amd/libfdoct_hip.so"
    [ -f "$lib" ] || { echo "missing $lib"; exit 1; }
    # shellcheck disable=SC2046
    env FDOCT_LIB="$lib" $(echo "$envs" | tr ',' ' ') python3 bench.py --steps ${AB_STEPS:-600} --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $round %-34s %.1f M A-scans/s  kernel %.4f ms  frac %.4f  parity %s' % ('$spec', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['parity'].get('worst_db_err_over_tol', d['parity'])))
"
  done
done
