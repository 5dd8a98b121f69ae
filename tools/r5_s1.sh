#!/bin/bash
# round-5 session 1: the new parity tests + the full GPU suite, device->host transfer forms, the streamed host-visible rollout,
# the interleaved MFMA / VALU microbench, config-4 stream A/B, the pad kernels.  Everything lands in gpurun_out/r5a.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_steps.py tests/test_gpu_kernels.py -m gpu -x -q -k "train_steps or pad2d" > $O/new_tests.log 2>&1; echo "new tests rc=$? $(tail -1 $O/new_tests.log)"
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_train_steps.py > $O/suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/suite.log)"
grep -E "FAILED|Error|error" $O/suite.log | head -20
timeout 300 python tools/bench_d2h.py > $O/d2h.json 2> $O/d2h.err; echo "d2h rc=$?"; cat $O/d2h.err | tail -40
for cfg in "dma 16 1" "dma 16 2" "kernel 8 1" "kernel 16 1" "kernel 32 1"; do
  set -- $cfg
  DLWP_D2H=$1 DLWP_D2H_BLOCKS=$2 DLWP_STREAMED_GROUPS=$3 timeout 300 python tools/bench_host_rollout.py --reps 4 > $O/host_$1_$2_$3.json 2> $O/host_$1_$2_$3.err
  echo "host $cfg: $(cut -c1-400 $O/host_$1_$2_$3.json)"; tail -2 $O/host_$1_$2_$3.err
done
timeout 120 tools/microbench/mfma_bf16_interleave.bin > $O/mfma_interleave.txt 2>&1; cat $O/mfma_interleave.txt
for cfg in "1 default" "0 default" "1 1" "1 split"; do
  set -- $cfg
  for m in 8 32; do
    if [ "$2" = default ]; then DLWP_ROLLOUT_OWN_STREAM=$1 timeout 300 python tools/bench_cfg4.py --members $m > $O/cfg4_own$1_$2_m$m.json 2> $O/cfg4.err
    else DLWP_ROLLOUT_OWN_STREAM=$1 DLWP_ROLLOUT_GROUPS=$2 timeout 300 python tools/bench_cfg4.py --members $m > $O/cfg4_own$1_$2_m$m.json 2> $O/cfg4.err; fi
    echo "cfg4 own=$1 groups=$2 m=$m: $(python -c "
import json,sys
d=json.load(open('$O/cfg4_own$1_$2_m$m.json'))
print(round(d['six_hour_steps_per_s']), round(d['ms_per_forward'],4))" 2>&1 | tail -1)"
  done
done
python tools/bench_pad.py --iters 20 > $O/pad.txt 2>&1; cat $O/pad.txt
