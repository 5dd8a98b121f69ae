#!/bin/bash
# knock-out builds of the channel-block weight gradient (conv_wgrad_cb_kernel.h: DLWP_WG_KNOCK = 1 no loads of the next tile inside
# the quad loop, 2 no MFMAs, 3 no LDS reads in the transforms, 4 no transforms) run through the phase-timing microbench.  Build here, run on the
# GPU box:  bash tools/knockout_wgrad_cb.sh build   |   gpurun -- 'bash tools/knockout_wgrad_cb.sh run'
cd "$(dirname "$0")/microbench"
if [ "$1" = build ]; then
  for k in 0 1 2 3 4; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDLWP_PHASE_TIMING -DDLWP_WG_KNOCK=$k -Wno-unused-result -o wgrad_cb_knock$k.bin \
      wgrad_cb_phase_timing.hip -I../../include || exit 1
  done
else
  for k in 0 1 2 3 4; do echo "== DLWP_WG_KNOCK=$k"; timeout 100 ./wgrad_cb_knock$k.bin; done
fi
