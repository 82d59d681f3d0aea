"""Copies the judged summaries of a tools/prof_c2.sh (+ prof_traffic.sh) run from gpurun_out/prof_<tag>/ into profiles/.
usage: python tools/collect_profiles.py <tag> <prefix>      e.g.  r01c r01_c2"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag, prefix = sys.argv[1], sys.argv[2]
src = "gpurun_out/prof_" + tag
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(src + "/kt/*/*_kernel_stats.csv")[0]
shutil.copy(ks, "profiles/%s_kernel_stats.csv" % prefix)
kt = glob.glob(src + "/kt/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(kt)) if "fused" in r["Kernel_Name"]]
keep = rows[:5] + rows[len(rows) // 2: len(rows) // 2 + 20] + rows[-5:]   # head, steady-state middle, tail
with open("profiles/%s_kernel_trace_fused.csv" % prefix, "w") as f:
    f.write("Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size_X,Workgroup_Size_X,VGPR_Count,Duration_us\n")
    for r in keep:
        f.write('"%s",%s,%s,%s,%s,%s,%.3f\n' % (r["Kernel_Name"], r["Start_Timestamp"], r["End_Timestamp"], r["Grid_Size_X"],
                                              r["Workgroup_Size_X"], r.get("VGPR_Count", ""),
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
with open("profiles/%s_pmc_summary.txt" % prefix, "w") as f:
    f.write(subprocess.check_output([sys.executable, "tools/prof_summary.py", src]).decode())
if os.path.exists(src + "/traffic.json"):
    t = json.load(open(src + "/traffic.json"))
    json.dump(t, open("profiles/%s_pmc_traffic.json" % prefix, "w"), indent=1)
    json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
print(open("profiles/%s_pmc_summary.txt" % prefix).read())
