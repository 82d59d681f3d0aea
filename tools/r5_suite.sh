#!/bin/bash
# the whole GPU suite (as the driver runs it) + the default bench line
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_tests.log
tail -8 gpurun_out/r5_tests.log
python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
echo "bench exit $?"
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_bench.json') if l.startswith('{')][0])
print(d['value'], d['roofline']['frac'], d['precise_division'])
print(d['parity'], d['cpu_baseline']['value'])
"
# the driver's own command next to it (K = 20, W = 5)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver.json 2>/dev/null
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_driver.json') if l.startswith('{')][0])
print('driver-sized run:', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
