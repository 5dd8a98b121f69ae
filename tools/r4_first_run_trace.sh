#!/bin/bash
OUT=gpurun_out/first/tr2
mkdir -p $OUT
DLWP_SEGV_TRACE=1 timeout 900 python -m pytest tests -m gpu -q -s -p no:faulthandler > $OUT/run.log 2>&1
echo "rc=$?" | tee $OUT/summary.txt
grep -n "native backtrace" -A45 $OUT/run.log | cut -c1-220 | tee -a $OUT/summary.txt | head -70
