#!/bin/bash
# Builds fdoct_amd/libfdoct_hip_<name>.so with extra -D flags on fdoct_kernels.hip (benchmark plan only), for tools/ab.sh.
# usage: tools/mkvariant.sh name [-DFLAG ...]
set -e
name=$1; shift
cd /root/repo/fdoct_amd/csrc
make -s fdoct_generic.o fdoct_display.o fdoct_host.o
mkdir -p /tmp/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DFDOCT_DEV_SINGLE "$@" \
  -c fdoct_kernels.hip -o /tmp/variants/k_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DFDOCT_DEV_SINGLE "$@" \
  -c fdoct_capi.cpp -o /tmp/variants/c_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfdoct_hip_$name.so /tmp/variants/k_$name.o /tmp/variants/c_$name.o fdoct_generic.o fdoct_display.o fdoct_host.o
echo built libfdoct_hip_$name.so
