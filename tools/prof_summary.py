"""Summarises a tools/prof_c2.sh output directory: kernel time and per-A-scan counter figures."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
rows_per_launch = None
for line in open(d + "/kt.log"):
    if line.startswith("{"):
        j = json.loads(line)
        rows_per_launch = j["roofline"]["ascans_per_launch"]
        print("bench: %.1f M A-scans/s, kernel %.4f ms, frac %.4f" % (j["value"] / 1e6, j["roofline"]["kernel_ms_avg"], j["roofline"]["frac"]))
for f in glob.glob(d + "/kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "fused" in r["Name"]:
            print("kernel-trace: calls %s avg %.1f us min %.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Name"][:80]))
agg = collections.defaultdict(list)
for f in glob.glob(d + "/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if rows_per_launch:
    for k in sorted(agg):
        v = sum(agg[k]) / len(agg[k])
        print("%-24s %14.0f  per A-scan %10.1f" % (k, v, v / rows_per_launch))
