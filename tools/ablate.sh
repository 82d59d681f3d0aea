#!/bin/bash
# Stage-cost ablation of the fused kernel (profiling aid; outputs are wrong while a bit is set).
# bits: 1 preprocess math, 2 staging+gather, 4/8/16 FFT pass 1/2/3, 32 untangle+magnitude, 64 log, 128 stores, 256 loads
cd "$GRAFT_REPO_ROOT"
for ab in 0 1 2 4 8 16 32 64 128 256 28 60 63 511; do
  FDOCT_ABLATE=$ab python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ablate %4d: %.1f M A-scans/s kernel %.4f ms' % ($ab, d['value']/1e6, d['roofline']['kernel_ms_avg']))
"
done
