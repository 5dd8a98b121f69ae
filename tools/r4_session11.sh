#!/bin/bash
OUT=gpurun_out/s11
mkdir -p $OUT
export TMPDIR=/tmp
DLWP_UNCACHED_FREE=1 timeout 300 python -X faulthandler tools/stress_grouped_rollout.py 300 > $OUT/stress_free.log 2>&1
echo "stress with hipFree per rollout: rc=$? $(tail -1 $OUT/stress_free.log | cut -c1-100)" | tee -a $OUT/summary.txt
timeout 300 python -X faulthandler tools/stress_grouped_rollout.py 300 > $OUT/stress_pool.log 2>&1
echo "stress with pooled uncached blocks: rc=$? $(tail -1 $OUT/stress_pool.log | cut -c1-100)" | tee -a $OUT/summary.txt
for i in 1 2 3; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/run_$i.log 2>&1
  rc=$?
  echo "suite run $i rc=$rc $(grep -E 'passed|failed' $OUT/run_$i.log | tail -1)" | tee -a $OUT/summary.txt
  if [ $rc -ne 0 ]; then grep -n "Fatal\|File \"/.*repo\|^FAILED" $OUT/run_$i.log | head -8 >> $OUT/summary.txt; else rm -f $OUT/run_$i.log; fi
done
python - <<'PY' >> $OUT/summary.txt
import json, time, torch, ctypes, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from dlwp_amd import ops
x = torch.randn(256, 4, 91, 180, device='cuda'); w = torch.randn(3, 3, 4, 32, device='cuda') * 0.1; b = torch.zeros(32, device='cuda')
cd = ops.make_conv(32, 3, 3, 2, ops.make_pad(2, 2, 2, 2, 0, 1), ops.ACT_TANH)
y = torch.empty(256, 32, 91, 180, device='cuda')
for mode in (0, 2):
    ops.set_few_stream(mode)
    for _ in range(5): ops.conv2d(x, w, b, cd, out=y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.conv2d(x, w, b, cd, out=y)
    torch.cuda.synchronize(); print('layer1 unpooled 91x180 x256, few_stream', mode, round((time.perf_counter() - t0) / 20 * 1e3, 4), 'ms')
ops.set_few_stream(1)
PY
cat $OUT/summary.txt
