#!/bin/bash
# control: the ROUND-3 tree's GPU suite as the first GPU process on a fresh box
OUT=$PWD/gpurun_out/first/r3
mkdir -p $OUT
cd _r3
timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/run.log 2>&1
rc=$?
echo "r3 tree first-run: rc=$rc $(grep -E 'passed|failed' $OUT/run.log | tail -1)" | tee $OUT/summary.txt
if [ $rc -ne 0 ]; then grep -n "Fatal\|File \"/.*repo\|^FAILED" $OUT/run.log | head -8 | tee -a $OUT/summary.txt; fi
