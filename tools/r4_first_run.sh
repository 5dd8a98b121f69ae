#!/bin/bash
# the full GPU suite as the FIRST GPU process on a fresh box (the driver's situation at round end); extra args: env assignments
TAG=$1; shift
OUT=gpurun_out/first/$TAG
mkdir -p $OUT
env "$@" timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/run.log 2>&1
rc=$?
echo "first-run trial $TAG ($*): rc=$rc $(grep -E 'passed|failed' $OUT/run.log | tail -1)" | tee $OUT/summary.txt
if [ $rc -ne 0 ]; then grep -n "Fatal\|File \"/.*repo\|^FAILED" $OUT/run.log | head -8 | tee -a $OUT/summary.txt; fi
