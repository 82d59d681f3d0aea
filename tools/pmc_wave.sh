#!/bin/bash
# SQ counters of the wave-per-row kernels on the shipped configurations (tools/bench_generic.py): two --pmc passes.
# gpurun -- 'bash tools/pmc_wave.sh'  ->  gpurun_out/pmc_wave/summary.txt (per launch and per input A-scan, by kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=gpurun_out/pmc_wave
rm -rf $d && mkdir -p $d
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $d/p1 -- python3 tools/bench_generic.py 0.05 > $d/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $d/p2 -- python3 tools/bench_generic.py 0.05 > $d/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/pmc_wave/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wave_kernel" not in k:
            continue
        k = re.sub(r"\(.*", "", k)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
# rows per launch of each shape in tools/bench_generic.py: nframes * H input A-scans
rows = {"<160, 4, 2560, unsigned char": 3490 * 120 // 1, "<640, 4, 2560, unsigned char": 0}
with open("gpurun_out/pmc_wave/summary.txt", "w") as o:
    for k in sorted(tot):
        o.write(k + "\n")
        waves = tot[k]["SQ_WAVES"] / max(1, n[k]["SQ_WAVES"])
        for c in sorted(tot[k]):
            v = tot[k][c] / n[k][c]
            o.write("   %-24s per launch %16.0f   per wave %12.1f\n" % (c, v, v / waves if waves else 0))
print(open("gpurun_out/pmc_wave/summary.txt").read())
PY
