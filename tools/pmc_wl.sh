#!/bin/bash
# SQ instruction counters of one bench workload (C1/C3/C4 ...): gpurun -- 'bash tools/pmc_wl.sh C3 C4'
# -> gpurun_out/pmc_wl/<wl>.txt : per-launch totals of the path's kernels and per-A-scan values.  PMC="..." picks other counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in "$@"; do
  d=gpurun_out/pmc_wl/$wl
  rm -rf $d && mkdir -p $d
  rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES} --output-format csv -d $d -- python3 bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 --workload $wl > $d/log.txt 2>&1
  python3 - "$d" "$wl" <<'PY'
import csv, glob, json, sys, collections
d, wl = sys.argv[1], sys.argv[2]
line = [l for l in open(d + "/log.txt") if l.startswith("{")][-1]
b = json.loads(line)
per_launch = b["roofline"]["ascans_per_launch"]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
        if not any(t in name for t in ("fused_kernel", "generic_kernel", "wave_kernel", "big_")):
            continue
        tot[(name, r["Counter_Name"])] += float(r["Counter_Value"]); n[(name, r["Counter_Name"])] += 1
with open("gpurun_out/pmc_wl/%s.txt" % wl, "w") as o:
    o.write("%s: %s\n" % (wl, b["config"]["workload"]))
    for k in sorted(tot):
        o.write("%-22s %-24s per launch %14.0f   per input A-scan %9.1f\n" % (k[0], k[1], tot[k] / n[k], tot[k] / n[k] / per_launch))
print(open("gpurun_out/pmc_wl/%s.txt" % wl).read())
PY
done
