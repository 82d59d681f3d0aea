cd "$GRAFT_REPO_ROOT"
for wk in "5 30" "100 300" "500 2000" "2000 5000" "5 30"; do set -- $wk; python3 bench.py --warmup $1 --steps $2 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('W=$1 K=$2  %.1f M A-scans/s  ms/step %.4f kernel %.4f ms frac %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
"; done
