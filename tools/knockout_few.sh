#!/bin/bash
# Profiling builds of the few-channel streaming kernel (csrc/conv_fwd_few.hip) with one phase removed (DLWP_KNOCK_FEW):
# dlwp_amd/knock/libdlwp_hip_few<n>.so, selected with DLWP_LIB_PATH.  Results are WRONG by construction.
cd "$(dirname "$0")/../dlwp_amd/csrc" || exit 1
make -j16 > /dev/null || exit 1
mkdir -p build/knock ../knock
for k in "$@"; do
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DDLWP_KNOCK_FEW=$k -c conv_fwd_few.hip -o build/knock/conv_fwd_few_$k.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../knock/libdlwp_hip_few$k.so $(ls build/*.o | grep -v "conv_fwd_few.o") build/knock/conv_fwd_few_$k.o ) &
done
wait
ls ../knock/
