#!/bin/bash
# rocprofv3 evidence of `bench.py --workload INI` (the shipped build/BscanFFT.ini shape on the wave-per-row kernel):
# kernel trace + two SQ-counter passes.  gpurun -- 'bash tools/prof_ini.sh r03'  then  python tools/ini_summary.py r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r03}
out=gpurun_out/prof_ini_$tag
rm -rf $out && mkdir -p $out
B="--workload INI --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py $B --steps 200 > $out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc1 -- python3 bench.py $B --steps 6 --warmup 2 --ramp-seconds 0 > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc2 -- python3 bench.py $B --steps 6 --warmup 2 --ramp-seconds 0 > $out/pmc2.log 2>&1
tail -c 300 $out/kt.log
