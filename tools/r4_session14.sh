#!/bin/bash
OUT=gpurun_out/s14
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "position_split" > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)" | tee $OUT/summary.txt
grep -n "^FAILED\|^E  " $OUT/tests.log | head -12 >> $OUT/summary.txt
echo "base: $(python tools/bench_layer6.py 2>/dev/null)" >> $OUT/summary.txt
for k in 1 2 4 8 16 32 10; do
  echo "knock $k: $(DLWP_LIB_PATH=$PWD/dlwp_amd/knock/libdlwp_hip_w2s$k.so python tools/bench_layer6.py --modes 1 2>/dev/null)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
