#!/bin/bash
# The counter passes of tools/prof_round.sh alone (SQ counters, HBM traffic; row-major and transposed), for a re-take after a change
# that touches what the counter runs launch.  usage: gpurun --timeout 900 -- 'bash tools/prof_counters.sh r06'
set -o pipefail
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
B="--steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0"
rm -rf $out/pmc1 $out/pmc2 $out/fetch $out/write $out/fetch_t $out/write_t
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc1 -- python3 bench.py $B > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc2 -- python3 bench.py $B > $out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py $B > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py $B > $out/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_t -- python3 bench.py $B --layout transposed > $out/fetch_t.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_t -- python3 bench.py $B --layout transposed > $out/write_t.log 2>&1
ls $out
