#!/bin/bash
# wave-per-row kernels after the pruned last pass: their tests, the INI bench line, the five shipped configurations
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_jit.py tests/test_gpu_parity.py -q -m gpu -x -k "wave or jit or shipped or generic or deep or dispersion" > gpurun_out/r5_prune_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_prune_tests.log
tail -3 gpurun_out/r5_prune_tests.log
python3 bench.py --workload INI --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 2>/dev/null | grep "^{" > gpurun_out/r5_prune_ini.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r5_prune_ini.json').read()); print('INI %.1f M input A-scans/s  %.4f ms  %s' % (d['value']/1e6, d['ms_per_step'], d.get('kernel','')))"
python3 tools/bench_generic.py 1.0 > gpurun_out/r5_prune_generic.txt 2>&1
cat gpurun_out/r5_prune_generic.txt
