"""Summary of a tools/prof_ini.sh run -> profiles/<tag>_ini_kernel_stats.csv, profiles/<tag>_ini_summary.txt and
profiles/pmc_ini.json (the per-A-scan on-chip figures bench.py --workload INI reports).   python tools/ini_summary.py r03"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/prof_ini_" + tag


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def bench_line(path):
    j = None
    for line in open(path, errors="replace"):
        if line.startswith("{"):
            j = json.loads(line)
    return j


j = bench_line(src + "/kt.log")
rows = j["roofline"]["ascans_per_launch"]
lines = ["command: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload INI --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --steps 200",
         "bench line: %.1f M input A-scans/s, step %.4f ms (HIP events), %d input A-scans per launch" % (j["value"] / 1e6, j["roofline"]["kernel_ms_avg"], rows)]
ks = newest(src + "/kt/**/*_kernel_stats.csv")
shutil.copy(ks, "profiles/%s_ini_kernel_stats.csv" % tag)
for r in csv.DictReader(open(ks)):
    if "wave_kernel" in r["Name"] or "bin2x2" in r["Name"]:
        lines.append("kernel-trace: calls %s avg %.1f us min %.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Name"][:90]))
agg = collections.defaultdict(list)
for f in glob.glob(src + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wave_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
per = {}
for k in sorted(agg):
    v = sum(agg[k]) / len(agg[k])
    per[k] = v / rows
    lines.append("%-24s per launch %16.0f   per input A-scan %10.1f" % (k, v, v / rows))
open("profiles/%s_ini_summary.txt" % tag, "w").write("\n".join(lines) + "\n")
json.dump({"tag": tag, "workload": "INI", "ascans_per_launch": rows,
           "lds_active_cycles_per_ascan": round(per.get("SQ_LDS_IDX_ACTIVE", 0.0), 1),
           "lds_bank_conflict_cycles_per_ascan": round(per.get("SQ_LDS_BANK_CONFLICT", 0.0), 1),
           "valu_insts_per_ascan": round(per.get("SQ_INSTS_VALU", 0.0), 1), "lds_insts_per_ascan": round(per.get("SQ_INSTS_LDS", 0.0), 1),
           "salu_insts_per_ascan": round(per.get("SQ_INSTS_SALU", 0.0), 1),
           "valu_active_cycles_per_ascan": round(per.get("SQ_ACTIVE_INST_VALU", 0.0), 1),
           "source": "rocprofv3 --pmc of `bench.py --workload INI` (tools/prof_ini.sh), wave_kernel launches, summed over the chip"},
          open("profiles/pmc_ini.json", "w"), indent=1)
print("\n".join(lines))
