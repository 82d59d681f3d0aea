#!/bin/bash
# rocprofv3 kernel-trace summaries of the non-headline bench workloads and modes -> gpurun_out/prof_workloads.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_workloads.txt
: > $out
run() {  # label, bench args...
  label=$1; shift
  # (WL="INI INI_generic" bash tools/prof_workloads.sh: only those legs)
  if [ -n "$WL" ] && ! echo " $WL " | grep -q " $label "; then return 0; fi
  d=gpurun_out/prof_wl_$label
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0.5 --steps 300 "$@" > $d.log 2>&1
  echo "== $label: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0.5 --steps 300 $*" >> $out
  grep '^{' $d.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   bench: %.1f M A-scans/s, %.4f ms per launch, %.0f GB/s algorithmic (%.1f %% of 8 TB/s), %.0f B per A-scan' % (d['value']/1e6, r['kernel_ms_avg'], r['achieved'], 100*r['frac'], r['algorithmic_bytes_per_ascan']))
p=d.get('roofline_power'); f=d.get('roofline_fp32')
if p: print('   power ceiling: %.0f W of %.0f W (%.1f %%), %.2f uJ per input A-scan = floor %.2f + HBM %.2f + on-chip %.2f; at the cap with this energy %.1f M A-scans/s' % (p['achieved'], p['peak'], 100*p['frac'], p['uj_per_ascan'], p['uj_floor'], p['uj_hbm'], p['uj_onchip'], p['ascans_per_s_at_the_cap_with_this_energy']/1e6))
if f: print('   fp32 ceiling: %.1f TFLOP/s of DFT arithmetic = %.1f %% of the 157.3 TFLOP/s vector peak' % (f['achieved'], 100*f['frac']))
for st in (d.get('stages') or []): print('   stage %-12s %.4f ms, %d B per input A-scan, %.0f GB/s = %.1f %% of 8 TB/s' % (st['stage'], st['kernel_ms_avg'], st['algorithmic_bytes_per_ascan'], st['achieved'], 100*st['frac']))" >> $out
  python3 - $d >> $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "fdoct::" in r["Name"]:
            print("   rocprofv3: calls %s avg %.1f us min %.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Name"][:110]))
PY
}
run C1 --workload C1
run C3 --workload C3
run C4 --workload C4
run C2_u8 --input-bits 8
run C2_bg2d --background-2d
run C2_transposed --layout transposed
run INI --workload INI --steps 100
run INI_generic --workload INI --steps 30 --plan -2
run C2_bg2d_transposed --background-2d --layout transposed
run C2_one_word --one-word-division
run C2_bg2d_one_word --background-2d --one-word-division
run C2_transposed_one_word --layout transposed --one-word-division
run C3_one_word --workload C3 --one-word-division
run C4_one_word --workload C4 --one-word-division
run LONG --workload LONG
run LONG4 --workload LONG4
cat $out
