#!/bin/bash
# rocprofv3 kernel trace of the staged (two-kernel) mode: gpurun -- 'bash tools/prof_staged.sh <tag>'
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --staged --steps 400 "$@" > $out/kt.log 2>&1
tail -1 $out/kt.log | cut -c1-200
head -4 $out/kt/*/*_kernel_stats.csv | cut -c1-200
