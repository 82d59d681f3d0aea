"""Copies the judged summaries of a tools/prof_round.sh run from gpurun_out/prof_<tag>/ into profiles/<tag>_*.
usage: python tools/collect_round.py r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = "gpurun_out/prof_" + tag
os.makedirs("profiles", exist_ok=True)
_glob = glob.glob


def newest(pattern):
    """gpurun merges a call's files into gpurun_out/ next to those of earlier calls: per directory keep the newest match."""
    best = {}
    for f in _glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


glob.glob = newest


def bench_line(path):
    j = None
    for line in open(path, errors="replace"):
        if line.startswith("{"):
            j = json.loads(line)
    return j


def stage_of(name):
    """fused_kernel<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, LEAN, STAGE, AVG, IB2D, NORM>"""
    if "fused_kernel" not in name:
        return None
    args = name[name.index("<") + 1:name.rindex(">")].split(",")
    return int(args[10])


# 1. kernel stats of the default command + a sample of dispatch rows of the fused chain
ks = glob.glob(src + "/kt/*/*_kernel_stats.csv")[0]
shutil.copy(ks, "profiles/%s_c2_kernel_stats.csv" % tag)
kt = glob.glob(src + "/kt/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(kt)) if stage_of(r["Kernel_Name"]) == 0]
if rows:  # only the full-chip launches of the timed configuration (bench.py's half-chip probe uses a smaller grid)
    full_grid = max(int(r["Grid_Size_X"]) for r in rows)
    rows = [r for r in rows if int(r["Grid_Size_X"]) == full_grid]
keep = rows[:5] + rows[len(rows) // 2: len(rows) // 2 + 20] + rows[-5:]
with open("profiles/%s_c2_kernel_trace_fused.csv" % tag, "w") as f:
    f.write("Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size_X,Workgroup_Size_X,VGPR_Count,Duration_us\n")
    for r in keep:
        f.write('"%s",%s,%s,%s,%s,%s,%.3f\n' % (r["Kernel_Name"], r["Start_Timestamp"], r["End_Timestamp"], r["Grid_Size_X"],
                                              r["Workgroup_Size_X"], r.get("VGPR_Count", ""),
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
j = bench_line(src + "/kt.log")
lines = []
lines.append("command: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline")
lines.append("bench line: %.1f M A-scans/s, kernel %.4f ms (HIP events in bench.py), roofline.frac %.4f" % (
    j["value"] / 1e6, j["roofline"]["kernel_ms_avg"], j["roofline"]["frac"]))
for s in j.get("stages") or []:
    lines.append("bench stage %-12s %.4f ms avg (%.4f min) over %d launches: %.0f GB/s = %.4f of 8 TB/s (%d B per A-scan)" % (
        s["stage"], s["kernel_ms_avg"], s["kernel_ms_min"], s["launches"], s["achieved"], s["frac"], s["algorithmic_bytes_per_ascan"]))
rows_per_launch = j["roofline"]["ascans_per_launch"]
for r in csv.DictReader(open(ks)):
    st = stage_of(r["Name"])
    if st is not None:
        nbytes = {0: 8192, 1: 12288, 2: 12288}[st]
        us = float(r["AverageNs"]) / 1e3
        lines.append("rocprofv3 %-22s calls %5s avg %.1f us min %.1f us -> %.0f GB/s algorithmic = %.4f of 8 TB/s   %s" % (
            {0: "fused chain", 1: "resample stage", 2: "FFT+mag+log stage"}[st], r["Calls"], us, float(r["MinNs"]) / 1e3,
            nbytes * rows_per_launch / us / 1e3, nbytes * rows_per_launch / us / 1e3 / 8000.0, r["Name"][:96]))
# 2. SQ counters per launch and per A-scan (fused chain only: the PMC runs use --stage-steps 0)
agg = collections.defaultdict(list)
for f in glob.glob(src + "/pmc[12]/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if stage_of(r["Kernel_Name"]) == 0:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
lines.append("")
lines.append("SQ counters of the fused chain (two --pmc passes, 10 launches each), per launch and per A-scan:")
for k in sorted(agg):
    v = sum(agg[k]) / len(agg[k])
    lines.append("%-24s %14.0f  per A-scan %10.1f" % (k, v, v / rows_per_launch))
open("profiles/%s_c2_pmc_summary.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))


# 3. HBM traffic
def avg(d, name):
    v = []
    for f in glob.glob(src + "/%s/*/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if stage_of(r["Kernel_Name"]) == 0 and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return sum(v) / len(v)


jf = bench_line(src + "/fetch.log")
fetch_kb, write_kb = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
res = {"tag": tag, "workload": jf["config"]["workload"].split(":")[0], "frames_per_step": jf["config"]["frames_per_step_per_gpu"],
       "ascans_per_launch": jf["roofline"]["ascans_per_launch"], "FETCH_SIZE_raw_KB": fetch_kb, "WRITE_SIZE_raw_KB": write_kb,
       # gfx950: FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced streaming read -> x2 (MI355X_MICROARCH.md);
       # WRITE_SIZE taken as reported (calibrated for 16 B/lane stores; ours are 4 B/lane, 256 B per wave instruction)
       "hbm_read_bytes_per_launch": fetch_kb * 1024 * 2, "hbm_write_bytes_per_launch": write_kb * 1024,
       "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
       "algorithmic_bytes_per_launch": jf["roofline"]["algorithmic_bytes_per_ascan"] * jf["roofline"]["ascans_per_launch"]}
json.dump(res, open("profiles/%s_c2_pmc_traffic.json" % tag, "w"), indent=1)
json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res))

# 4. other workloads
with open("profiles/%s_other_workloads_kt.txt" % tag, "w") as f:
    for wl, args in (("C1", "--workload C1"), ("C3", "--workload C3"), ("C4", "--workload C4"), ("C2u8", "--input-bits 8"),
                     ("C2bg2d", "--background-2d")):
        d = bench_line("%s/wl_%s.log" % (src, wl))
        if not d:
            continue
        r = d["roofline"]
        f.write("== %s: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 300 %s\n" % (wl, args))
        f.write("   bench: %.1f M A-scans/s, %.4f ms per launch, %.0f GB/s algorithmic (%.1f %% of 8 TB/s), %.0f B per A-scan, parity %s\n" % (
            d["value"] / 1e6, r["kernel_ms_avg"], r["achieved"], 100 * r["frac"], r["algorithmic_bytes_per_ascan"],
            d["parity"].get("worst_db_err_over_tol")))
        for s in d.get("stages") or []:
            f.write("   stage %-12s %.4f ms: %.0f GB/s = %.1f %% of 8 TB/s\n" % (s["stage"], s["kernel_ms_avg"], s["achieved"], 100 * s["frac"]))
        for kf in glob.glob("%s/wl_%s/*/*_kernel_stats.csv" % (src, wl)):
            for rr in csv.DictReader(open(kf)):
                if "fdoct::" in rr["Name"]:
                    f.write("   rocprofv3: calls %s avg %.1f us min %.1f us  %s\n" % (rr["Calls"], float(rr["AverageNs"]) / 1e3,
                                                                                   float(rr["MinNs"]) / 1e3, rr["Name"][:110]))
print(open("profiles/%s_other_workloads_kt.txt" % tag).read())

# 5. shipped ini configurations
with open("profiles/%s_shipped_ini.txt" % tag, "w") as f:
    f.write("command: rocprofv3 --kernel-trace --stats -- python3 tools/bench_generic.py   (raw camera frames in, software binning on\n"
            "the GPU, 10 averages, dB B-scans out; one line per build/*.ini of the reference)\n")
    for line in open(src + "/ini.log", errors="replace"):
        if "A-scans/s" in line:
            f.write(line)
    f.write("\nrocprofv3 kernel stats of the same run:\n")
    for kf in glob.glob(src + "/ini/*/*_kernel_stats.csv"):
        for rr in csv.DictReader(open(kf)):
            if "fdoct::" in rr["Name"]:
                f.write("   calls %5s avg %9.1f us min %9.1f us total %6.1f ms  %s\n" % (rr["Calls"], float(rr["AverageNs"]) / 1e3, float(rr["MinNs"]) / 1e3,
                                                                                       float(rr["TotalDurationNs"]) / 1e6, rr["Name"][:100]))
print(open("profiles/%s_shipped_ini.txt" % tag).read())
shutil.copy(src + "/inst_cost.txt", "profiles/%s_inst_cost.txt" % tag)
