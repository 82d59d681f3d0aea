#!/bin/bash
# HBM traffic of the bench kernel from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE do not fit one pass).  Writes gpurun_out/prof_<tag>/traffic.json; copy it
# to profiles/pmc_traffic.json for bench.py to report.   gpurun -- 'bash tools/prof_traffic.sh <tag> [bench args]'
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --precise-steps 0 "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --precise-steps 0 "$@" > $out/write.log 2>&1
python3 - $out "$@" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def avg(d, name):
    v = []
    for f in glob.glob(out + "/%s/*/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "fused" in r["Kernel_Name"] and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return sum(v) / len(v) if v else None
j = None
for line in open(out + "/fetch.log"):
    if line.startswith("{"):
        j = json.loads(line)
fetch_kb, write_kb = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
res = {"workload": j["config"]["workload"].split(":")[0], "frames_per_step": j["config"]["frames_per_step_per_gpu"],
       "ascans_per_launch": j["roofline"]["ascans_per_launch"],
       "FETCH_SIZE_raw_KB": fetch_kb, "WRITE_SIZE_raw_KB": write_kb,
       # gfx950: FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced streaming read -> x2;
       # WRITE_SIZE is taken as reported (calibrated for 16 B/lane stores; ours are 4 B/lane, 256 B per wave
       # instruction: uncalibrated, stated as such)
       "hbm_read_bytes_per_launch": fetch_kb * 1024 * 2, "hbm_write_bytes_per_launch": write_kb * 1024,
       "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
       "algorithmic_bytes_per_launch": j["roofline"]["algorithmic_bytes_per_ascan"] * j["roofline"]["ascans_per_launch"]}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
