#!/bin/bash
# All the rocprofv3 evidence of one round in one GPU call:  gpurun --timeout 1100 -- 'bash tools/prof_round.sh r02'
# Writes gpurun_out/prof_<tag>/...; tools/collect_round.py <tag> copies the judged summaries into profiles/<tag>_*.
# (--pmc passes are separate runs without any tracing flag besides what rocprofv3 needs; the program itself follows `--`.)
set -o pipefail
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
echo "[1/7] kernel trace of the default bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 > $out/kt.log 2>&1
echo "[2/7] SQ counters, pass 1"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc1 -- python3 bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 > $out/pmc1.log 2>&1
echo "[3/7] SQ counters, pass 2"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc2 -- python3 bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 > $out/pmc2.log 2>&1
echo "[4/7] HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 > $out/write.log 2>&1
echo "[5/7] other workloads"
for wl in C1 C3 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/wl_$wl -- python3 bench.py --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --steps 300 --workload $wl > $out/wl_$wl.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/wl_C2u8 -- python3 bench.py --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --steps 300 --input-bits 8 > $out/wl_C2u8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/wl_C2bg2d -- python3 bench.py --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --steps 300 --background-2d > $out/wl_C2bg2d.log 2>&1
echo "[6/7] shipped ini configurations (wave-per-row kernels)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ini -- python3 tools/bench_generic.py > $out/ini.log 2>&1
echo "[7/7] instruction costs"
[ -x tools/ubench/inst_cost ] || hipcc -O3 --offload-arch=gfx950 -o tools/ubench/inst_cost tools/ubench/inst_cost.hip
timeout -k 10 120 tools/ubench/inst_cost > $out/inst_cost.txt 2>&1
tail -3 $out/kt.log | cut -c1-300
grep -v amdgpu.ids $out/ini.log
