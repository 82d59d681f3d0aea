for wl in C1 C3 C4; do for lay in rowmajor transposed; do python3 bench.py --workload $wl --layout $lay --steps 300 --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl %-10s %.1f M A-scans/s  %.4f ms  frac %.3f' % ('$lay', d['value']/1e6, d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
"; done; done
