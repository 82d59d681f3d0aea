#!/bin/bash
# the driver's command (K = 20, W = 5) with the ramp as bursts (round 4) and continuous (round 5), and K = 1000, interleaved
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r5_ramp.txt
: > $out
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1: %.1f M A-scans/s  %.4f ms  frac %.4f  sclk %s  %s W' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], (d.get('power') or {}).get('sclk_mhz_avg'), (d.get('power') or {}).get('package_w_avg')))
" >> $out; }
for r in 1 2 3; do
  FDOCT_BENCH_RAMP=bursts python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "K=20 bursts    "
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "K=20 continuous"
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --ramp-seconds 1.0 --no-cpu-baseline 2>/dev/null | line "K=20 cont. 1 s "
  python3 bench.py --no-cpu-baseline 2>/dev/null | line "K=1000         "
done
cat $out
