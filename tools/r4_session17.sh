#!/bin/bash
OUT=gpurun_out/s17
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_train_fold.py tests/test_gpu_parallel.py tests/test_gpu_configs.py -q -x > $OUT/tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' $OUT/tests.log | tail -1)" | tee $OUT/summary.txt
grep -n "^FAILED\|^E  " $OUT/tests.log | head -12 >> $OUT/summary.txt
for b in 8 4 16; do for p in 0 1; do
  echo "batch $b pair=$p: $(DLWP_TRAIN_PAIR=$p timeout 300 python tools/bench_train.py --batch $b --steps 60 --warmup 20 2>/dev/null | tail -1 | cut -c80-260)" >> $OUT/summary.txt
done; done
cat $OUT/summary.txt
