#!/bin/bash
# A/B of library variants on one GPU box, interleaved rounds (rule: compare within one process/box).
# usage (through gpurun): bash tools/ab.sh variantA variantB ...   (fdoct_amd/libfdoct_hip_<variant>.so)
cd "$GRAFT_REPO_ROOT"
cp fdoct_amd/libfdoct_hip.so /tmp/orig.so
for round in 1 2 3; do
  for v in "$@"; do
    cp fdoct_amd/libfdoct_hip_$v.so fdoct_amd/libfdoct_hip.so
    python3 bench.py --steps 600 --warmup 20 --no-cpu-baseline $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $round %-8s %.1f M A-scans/s  kernel %.4f ms  parity %s' % ('$v', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['parity']))
"
  done
done
cp /tmp/orig.so fdoct_amd/libfdoct_hip.so
