#!/bin/bash
# the wave-per-row kernels' evidence again after they changed: shipped configurations under rocprofv3, the INI legs of prof_workloads.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r05
mkdir -p $out && rm -rf $out/ini
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ini -- python3 tools/bench_generic.py > $out/ini.log 2>&1
grep "A-scans/s" $out/ini.log | cut -c1-150
WL="INI INI_generic" bash tools/prof_workloads.sh
cat gpurun_out/prof_workloads.txt | cut -c1-200
