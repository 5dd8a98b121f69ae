#!/bin/bash
# Profiling builds of the direct fp32 family (3x3 dilation 2 = layer 1 of the U-Net) with one phase removed (conv_fwd_kernel.h:
# DLWP_KNOCK_F32): dlwp_amd/knock/libdlwp_hip_f<n>.so, selected with DLWP_LIB_PATH.  Results are WRONG by construction.
cd "$(dirname "$0")/../dlwp_amd/csrc" || exit 1
make -j16 > /dev/null || exit 1
mkdir -p build/knock ../knock
for k in "$@"; do
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 -DDLWP_KNOCK_F32=$k -c conv_fwd_k3d2.hip -o build/knock/conv_fwd_k3d2_f$k.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../knock/libdlwp_hip_f$k.so $(ls build/*.o | grep -v "conv_fwd_k3d2.o") build/knock/conv_fwd_k3d2_f$k.o ) &
done
wait
ls ../knock/
