#!/bin/bash
# Profiling builds of the streaming position-split Winograd kernel (csrc/conv_fwd_wino2s.hip) with one phase removed
# (DLWP_KNOCK_W2S): dlwp_amd/knock/libdlwp_hip_w2s<n>.so, selected with DLWP_LIB_PATH.  Results are WRONG by construction.
cd "$(dirname "$0")/../dlwp_amd/csrc" || exit 1
make -j16 > /dev/null || exit 1
mkdir -p build/knock ../knock
for k in "$@"; do
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DDLWP_KNOCK_W2S=$k -c conv_fwd_wino2s.hip -o build/knock/conv_fwd_wino2s_$k.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../knock/libdlwp_hip_w2s$k.so $(ls build/*.o | grep -v "conv_fwd_wino2s.o") build/knock/conv_fwd_wino2s_$k.o ) &
done
wait
ls ../knock/
