#!/bin/bash
# Profiling builds of the bf16 family with one phase of the octet cell-update kernels removed (conv_fwd_bf16_kernel.h:
# DLWP_KNOCK): dlwp_amd/knock/libdlwp_hip_k<n>.so, selected with DLWP_LIB_PATH.  Results are WRONG by construction: timing only.
cd "$(dirname "$0")/../dlwp_amd/csrc" || exit 1
make -j16 > /dev/null || exit 1
mkdir -p build/knock ../knock
for k in "$@"; do
  # (both translation units of the family: conv_fwd_bf16.hip and the octet instances of conv_fwd_bf16_o8.hip)
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DDLWP_KNOCK=$k -c conv_fwd_bf16.hip -o build/knock/conv_fwd_bf16_k$k.o &&
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DDLWP_KNOCK=$k -c conv_fwd_bf16_o8.hip -o build/knock/conv_fwd_bf16_o8_k$k.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../knock/libdlwp_hip_k$k.so $(ls build/*.o | grep -v -e 'build/conv_fwd_bf16\.o' -e 'build/conv_fwd_bf16_o8\.o') \
          build/knock/conv_fwd_bf16_k$k.o build/knock/conv_fwd_bf16_o8_k$k.o ) &
done
wait
ls -la ../knock/
