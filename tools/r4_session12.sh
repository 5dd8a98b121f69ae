#!/bin/bash
OUT=gpurun_out/s12
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/run.log 2>&1
echo "suite rc=$? $(grep -E 'passed|failed' $OUT/run.log | tail -1)" | tee $OUT/summary.txt
grep -n "Fatal\|File \"/.*repo\|^FAILED\|^E  " $OUT/run.log | head -12 >> $OUT/summary.txt
cp gpurun_out/forward_errors.json $OUT/ 2>/dev/null
bash tools/r3_trace.sh t8 8 > $OUT/trace_b8.txt 2>&1
cp gpurun_out/r3t/t8.json $OUT/ 2>/dev/null
tail -45 $OUT/trace_b8.txt >> $OUT/summary.txt
cat $OUT/summary.txt
