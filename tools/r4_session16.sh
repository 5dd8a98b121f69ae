#!/bin/bash
OUT=gpurun_out/s16
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pair.py -q -x > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)" | tee $OUT/summary.txt
grep -n "^FAILED\|^E  " $OUT/tests.log | head -12 >> $OUT/summary.txt
for b in 8 4 16; do echo "batch $b: $(timeout 300 python tools/bench_pair.py --batch $b 2>&1 | tail -1)" >> $OUT/summary.txt; done
cat $OUT/summary.txt
