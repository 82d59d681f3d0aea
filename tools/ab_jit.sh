#!/bin/bash
# A/B of compile-time options of the RUN-TIME compiled wave-per-row kernels on one GPU box, interleaved rounds.
# usage (through gpurun): bash tools/ab_jit.sh "<defines A>" "<defines B>" ...   e.g.  "" "-DFDOCT_WAVE_RESGI=0"
# Each variant runs `bench.py --workload INI` (AB_ARGS overrides) with FDOCT_JIT_DEFINES set and a cache directory of its own.
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root" || exit 1
rounds=${AB_ROUNDS:-3}
for round in $(seq 1 "$rounds"); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    FDOCT_JIT_DEFINES="$v" FDOCT_JIT_CACHE=/tmp/abjit_$i python3 bench.py ${AB_ARGS:---workload INI} --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('round $round [%-28s] %.1f M A-scans/s  %.4f ms  parity %s' % ('$v', d['value']/1e6, d['ms_per_step'], d['parity'].get('worst_db_err_over_tol', d['parity'])))
"
  done
done
