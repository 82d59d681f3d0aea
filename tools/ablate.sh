#!/bin/bash
# Stage-cost ablation of the fused kernel (profiling aid; outputs are wrong while a bit is set).
# needs a library built with -DFDOCT_RUNTIME_ABLATE (tools/mkvariant.sh abl -DFDOCT_RUNTIME_ABLATE)
# bits: 1 preprocess math, 2 staging+gather, 4 FFT, 32 untangle+magnitude, 64 log, 128 stores, 256 loads
cd "$GRAFT_REPO_ROOT"
for ab in ${ABL_SET:-0 1 2 4 32 64 128 256 384 63 511}; do
  FDOCT_ABLATE=$ab python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ablate %4d: %.1f M A-scans/s kernel %.4f ms' % ($ab, d['value']/1e6, d['roofline']['kernel_ms_avg']))
"
done
