#!/bin/bash
# The few-channel streaming kernel (csrc/conv_fwd_few.hip) on layer 1 of the config-2 U-Net: A/B against the general instance at
# several ensemble sizes, the knock-out builds (tools/knockout_few.sh <masks> must have been run HERE first: the libraries travel
# with the snapshot) and the s_memtime / s_memrealtime stamps (tools/microbench/few_phase_timing.bin, built here as well).
# usage (GPU box): bash tools/profile_few.sh r3   -> gpurun_out/prof/<tag>_few_stream.txt
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof; mkdir -p $O
cd $R
{
  echo "kernel source $(python -c 'from dlwp_amd import _lib; print(_lib.kernel_source_hash())')"
  echo "== layer 1 alone (4 -> 32, 3x3 dilation 2, zero rows / periodic columns, tanh, pooled epilogue, 88 x 180), tools/bench_layer1.py: ms per launch"
  for n in 16 24 32 64 128 256 1024; do
    a=$(DLWP_FEW_STREAM=0 python tools/bench_layer1.py --members $n --iters 50 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['pooled_88x180']['ms'])")
    b=$(DLWP_FEW_STREAM=2 python tools/bench_layer1.py --members $n --iters 50 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['pooled_88x180']['ms'])")
    echo "   $n members: general instance $a | streaming kernel $b"
  done
  echo "== knock-out builds at 256 members (DLWP_KNOCK_FEW bit mask: 1 barrier, 2 activation, 4 stores, 8 matrix loop, 16 loads; results wrong by construction)"
  for k in 1 2 4 8 16 20 12 24 28; do
    f=$R/dlwp_amd/knock/libdlwp_hip_few$k.so
    [ -f $f ] || continue
    m=$(DLWP_LIB_PATH=$f python tools/bench_layer1.py --members 256 --iters 50 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['pooled_88x180']['ms'])")
    echo "   without mask $k: $m ms"
  done
  echo "== stamps of wave 0 (tools/microbench/few_phase_timing.hip), 256 members; grid = workgroups"
  for g in 512 768 1024; do tools/microbench/few_phase_timing.bin 256 $g; done
} > $O/${TAG}_few_stream.txt 2>&1
cat $O/${TAG}_few_stream.txt | cut -c1-250
