#!/bin/bash
# Package power and sclk while the whole chip runs one instruction kind (tools/ubench/power_probe): energy per instruction.
# gpurun -- 'bash tools/power_probe.sh'  ->  gpurun_out/power_probe.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/power_probe.txt
: > $out
[ tools/ubench/power_probe -nt tools/ubench/power_probe.hip ] || hipcc -O3 --offload-arch=gfx950 -o tools/ubench/power_probe tools/ubench/power_probe.hip
for k in ${KINDS:-nop add fma pk_add pk_mul pk_fma pk_fma_32 pk_fma_8 pk_fma_1 sqrt ds_read_b64 ds_write_b64}; do
  timeout -k 5 30 tools/ubench/power_probe $k 6 > /tmp/pp_$k.txt 2>&1 &
  bp=$!
  sleep 3
  s=$(rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|Package Power" | sed 's/GPU\[0\]//; s/\t//g' | tr '\n' ' ')
  wait $bp
  echo "$(cat /tmp/pp_$k.txt) | $s" | tee -a $out
done
