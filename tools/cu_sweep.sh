#!/bin/bash
# The fused C2 chain on a part of the chip (grid of 32 ... 256 workgroups, one per CU): per-CU rate, package power and sclk.
# If the full-chip rate is set by the package power cap, the per-CU rate rises when fewer CUs run (higher clock, power under the cap).
# gpurun -- 'WL=C4 bash tools/cu_sweep.sh'  ->  gpurun_out/cu_sweep_<workload>.txt   (default workload C2)
cd "$GRAFT_REPO_ROOT"
wl=${WL:-C2}
out=gpurun_out/cu_sweep_$wl.txt
echo "workload $wl";  echo "workgroups | A-scans/s | per workgroup | package W (second half of the region) | sclk MHz avg" > $out
for b in 32 64 96 128 160 192 224 256; do
  python3 bench.py --workload $wl --blocks $b --steps 400 --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['power'] or {}
        print('%4d | %.1f M | %.3f M | %s | %s' % ($b, d['value']/1e6, d['value']/1e6/$b, p.get('package_w_last_half'), p.get('sclk_mhz_avg')))
" | tee -a $out
done
