#!/bin/bash
# SQ / store-path counters of the path's kernels through rocprofv3 --pmc (counters only: no trace domains next to --pmc on this
# pool) -- ONE script for what used to be pmc_wl.sh, pmc_wave.sh and pmc_store.sh.
# usage (through gpurun):
#   bash tools/pmc.sh wl C3 C4 ...     instruction counters of bench workloads, per launch and per input A-scan
#                                      -> gpurun_out/pmc_wl/<WL>.txt       (PMC="..." picks other counters; BARGS="..." adds bench arguments)
#   bash tools/pmc.sh wave             the wave-per-row kernels on the shipped configurations (tools/bench_generic.py), two passes
#                                      -> gpurun_out/pmc_wave/summary.txt
#   bash tools/pmc.sh store [variant]  store-path counters, row-major against the transposed store
#                                      -> gpurun_out/pmc_store/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mode=$1; shift
BENCH="--steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0"
INST="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES"
BUSY="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
summarise() {  # dir, output file, title, divisor (0: per wave from SQ_WAVES), kernel-name substrings...
  python3 - "$@" <<'PY'
import csv, glob, collections, re, sys
d, outp, title, div = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
keys = sys.argv[5:]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        if not any(t in name for t in keys):
            continue
        short = name if div == 0 else name.split("<")[0].split("::")[-1]
        tot[(short, r["Counter_Name"])] += float(r["Counter_Value"]); n[(short, r["Counter_Name"])] += 1
with open(outp, "a") as o:
    o.write(title + "\n")
    for k in sorted(tot):
        v = tot[k] / n[k]
        if div:
            o.write("%-22s %-26s per launch %16.0f   per input A-scan %10.2f\n" % (k[0], k[1], v, v / div))
        else:
            w = tot.get((k[0], "SQ_WAVES"), 0) / max(1, n.get((k[0], "SQ_WAVES"), 1))
            o.write("%s\n   %-26s per launch %16.0f   per wave %12.1f\n" % (k[0], k[1], v, v / w if w else 0))
print(open(outp).read())
PY
}
case $mode in
  wl)
    for wl in "$@"; do
      d=gpurun_out/pmc_wl/$wl; rm -rf "$d" gpurun_out/pmc_wl/$wl.txt; mkdir -p "$d"
      # shellcheck disable=SC2086
      timeout -k 10 300 rocprofv3 --pmc ${PMC:-$INST} --output-format csv -d "$d" -- python3 bench.py $BENCH $BARGS --workload "$wl" > "$d/log.txt" 2>&1
      per=$(python3 -c "import json,sys; print(json.loads([l for l in open('$d/log.txt') if l.startswith('{')][-1])['roofline']['ascans_per_launch'])")
      summarise "$d" "gpurun_out/pmc_wl/$wl.txt" "$wl $BARGS" "$per" fused_kernel generic_kernel wave_kernel big_
    done ;;
  wave)
    d=gpurun_out/pmc_wave; rm -rf $d; mkdir -p $d
    # shellcheck disable=SC2086
    timeout -k 10 300 rocprofv3 --pmc $INST --output-format csv -d $d/p1 -- python3 tools/bench_generic.py 0.05 > $d/p1.log 2>&1 &&
    timeout -k 10 300 rocprofv3 --pmc $BUSY --output-format csv -d $d/p2 -- python3 tools/bench_generic.py 0.05 > $d/p2.log 2>&1
    summarise $d $d/summary.txt "wave-per-row kernels, shipped configurations" 0 wave_kernel ;;
  store)
    v=${1:-base}; lib="$GRAFT_REPO_ROOT/fdoct_amd/libfdoct_hip_$v.so"; [ "$v" = base ] && lib="$GRAFT_REPO_ROOT/fdoct_amd/libfdoct_hip.so"
    export FDOCT_LIB="$lib"
    d=gpurun_out/pmc_store; rm -rf $d; mkdir -p $d
    P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM"
    # round 6: what backs up between the SQ and the texture addresser when the chain writes D x H itself -- the SQ's own view of the
    # vector-memory path (FIFO-full cycles towards the TA: write data, addresses, commands), a counter under DESIGN's "the CU's
    # store path for 64-byte segments" 
    P2="SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"
    for layout in rowmajor transposed; do   # (TCC_* passes abort or do not come back on this pool: left out; TA_* only with TA=1, last)
      # shellcheck disable=SC2086
      timeout -k 10 150 rocprofv3 --pmc $P1 --output-format csv -d $d/$layout -- python3 bench.py $BENCH --layout $layout > $d/$layout.log 2>&1 || break
      summarise $d/$layout $d/summary.txt "== $layout" 262000 fused_kernel
      # shellcheck disable=SC2086
      timeout -k 10 150 rocprofv3 --pmc $P2 --output-format csv -d $d/${layout}_fifo -- python3 bench.py $BENCH --layout $layout > $d/${layout}_fifo.log 2>&1 || break
      summarise $d/${layout}_fifo $d/summary.txt "== $layout, SQ -> TA FIFO counters" 262000 fused_kernel
    done
    if [ "$TA" = 1 ]; then
      for layout in rowmajor transposed; do
        timeout -k 10 120 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WRITE_WAVEFRONTS_sum --output-format csv -d $d/${layout}_ta -- python3 bench.py $BENCH --layout $layout > $d/${layout}_ta.log 2>&1 || break
        summarise $d/${layout}_ta $d/summary.txt "== $layout, TA counters" 262000 fused_kernel
      done
    fi ;;
  *) echo "usage: pmc.sh wl <workload...> | wave | store [variant]"; exit 1 ;;
esac
