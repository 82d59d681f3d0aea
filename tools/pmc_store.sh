#!/bin/bash
# Store-path counters of the fused chain, row-major against the transposed store: two --pmc passes per mode
# (SQ and L2-side counters; a pass of TA_* / TCP_* sums did not come back within 7 minutes on this pool and is left out).
# usage (gpurun): bash tools/pmc_store.sh [lib-variant]   -> gpurun_out/pmc_store/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
v=${1:-base}
lib="$GRAFT_REPO_ROOT/fdoct_amd/libfdoct_hip_$v.so"; [ "$v" = base ] && lib="$GRAFT_REPO_ROOT/fdoct_amd/libfdoct_hip.so"
export FDOCT_LIB="$lib"
d=gpurun_out/pmc_store
rm -rf $d && mkdir -p $d
B="--steps 6 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM"
P3="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_REQ_sum TCC_HIT_sum"
for mode in rowmajor transposed; do
  i=0
  for P in "$P1"; do   # (the TCC_* pass aborts inside rocprofv3 on this pool: left out)
    i=$((i+1))
    echo "$mode pass $i"
    timeout -k 10 150 rocprofv3 --pmc $P --output-format csv -d $d/${mode}_p$i -- python3 bench.py $B --layout $mode > $d/${mode}_p$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
out = open("gpurun_out/pmc_store/summary.txt", "w")
for mode in ("rowmajor", "transposed"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_store/%s_p*/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fused_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.write("== %s (per launch of 262000 A-scans; per A-scan)\n" % mode)
    for k in sorted(agg):
        v = sum(agg[k]) / len(agg[k])
        out.write("   %-36s %16.0f %12.2f\n" % (k, v, v / 262000))
out.close()
print(open("gpurun_out/pmc_store/summary.txt").read())
PY
