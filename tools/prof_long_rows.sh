#!/bin/bash
# Rows beyond the two-buffer LDS kernels: bench.py --workload LONG (4096 samples x8 -> 32768 points) and
# LONG4 (x4 -> 16384) on the route the library takes, on the long-row path (FDOCT_FORCE_LONG_ROWS=1: grouped launches) and on
# round 3's form of it (FDOCT_BIG_PER_PASS=1), then the SQ counters of the default route of LONG.
# gpurun -- 'bash tools/prof_long_rows.sh'  ->  gpurun_out/long_rows.txt
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/long_rows.txt; : > $out
B="--no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0"
line() {  # label, env..., workload
  local label=$1; shift; local wl=${!#}; set -- "${@:1:$(($#-1))}"
  echo "== --workload $wl $label" >> $out
  env "$@" python3 bench.py $B --workload $wl 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.3e A-scans/s, %.3f ms per step of %d A-scans, kernel(s): %s, parity %s of the tolerance' % (d['value'], d['ms_per_step'], d['roofline']['ascans_per_launch'], d['roofline']['kernel'], d['parity']['worst_db_err_over_tol']))" >> $out
}
line "(default route)" X=1 LONG
line "FDOCT_FORCE_LONG_ROWS=1 (grouped launches)" FDOCT_FORCE_LONG_ROWS=1 LONG
line "FDOCT_FORCE_LONG_ROWS=1 FDOCT_BIG_PER_PASS=1 (round 3: one launch per pass)" FDOCT_FORCE_LONG_ROWS=1 FDOCT_BIG_PER_PASS=1 LONG
line "(default route)" X=1 LONG4
line "FDOCT_GENERIC_INPLACE_ABOVE=81920 (one buffer in place)" FDOCT_GENERIC_INPLACE_ABOVE=81920 LONG4
line "FDOCT_FORCE_LONG_ROWS=1 (grouped launches)" FDOCT_FORCE_LONG_ROWS=1 LONG4
line "FDOCT_FORCE_LONG_ROWS=1 FDOCT_BIG_PER_PASS=1 (round 3: one launch per pass)" FDOCT_FORCE_LONG_ROWS=1 FDOCT_BIG_PER_PASS=1 LONG4
echo "== SQ counters of the default route of LONG (two --pmc passes; cycles are summed over CUs / SEs as rocprofv3 reports them)" >> $out
PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" bash tools/pmc.sh wl LONG > /dev/null 2>&1 && tail -n +2 gpurun_out/pmc_wl/LONG.txt >> $out
PMC="SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" bash tools/pmc.sh wl LONG > /dev/null 2>&1 && tail -n +2 gpurun_out/pmc_wl/LONG.txt >> $out
cat $out
