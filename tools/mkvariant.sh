#!/bin/bash
# Builds fdoct_amd/libfdoct_hip_<name>.so with extra -D flags on fdoct_kernels.hip (benchmark plan only), for tools/ab.sh.
# usage: tools/mkvariant.sh name [-DFLAG ...]      (PLAN_FLAGS overrides the default -DFDOCT_DEV_SINGLE)
set -e
name=$1; shift
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$root/fdoct_amd/csrc"
make -s fdoct_generic.o fdoct_wave.o fdoct_wave_x1.o fdoct_wave_x2.o fdoct_big.o fdoct_display.o fdoct_host.o fdoct_jit.o
tmp=${TMPDIR:-/tmp}/variants; mkdir -p "$tmp"
flags="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off ${PLAN_FLAGS--DFDOCT_DEV_SINGLE}"
hipcc --offload-arch=gfx950 $flags "$@" -c fdoct_kernels.hip -o "$tmp/k_$name.o" &
for u in capi state route; do hipcc --offload-arch=gfx950 $flags "$@" -c fdoct_$u.cpp -o "$tmp/${u}_$name.o" & done
wobjs="fdoct_wave.o fdoct_wave_x1.o fdoct_wave_x2.o fdoct_jit.o"
if [ -n "$WAVE" ]; then  # WAVE=1: the flags reach the wave-per-row kernels, their host side and the embedded run-time source too
  hipcc --offload-arch=gfx950 $flags "$@" -c fdoct_wave.hip -o "$tmp/w0_$name.o" &
  hipcc --offload-arch=gfx950 $flags "$@" -DFDOCT_WAVE_EXTRA_TU=1 -c fdoct_wave.hip -o "$tmp/w1_$name.o" &
  hipcc --offload-arch=gfx950 $flags "$@" -DFDOCT_WAVE_EXTRA_TU=2 -c fdoct_wave.hip -o "$tmp/w2_$name.o" &
  hipcc --offload-arch=gfx950 $flags "$@" -c fdoct_jit.cpp -o "$tmp/j_$name.o" &
  wobjs="$tmp/w0_$name.o $tmp/w1_$name.o $tmp/w2_$name.o $tmp/j_$name.o"
fi
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfdoct_hip_$name.so "$tmp/k_$name.o" "$tmp/capi_$name.o" "$tmp/state_$name.o" "$tmp/route_$name.o" fdoct_generic.o $wobjs fdoct_big.o fdoct_display.o fdoct_host.o -ldl
echo built libfdoct_hip_$name.so
