#!/bin/bash
OUT=gpurun_out/s9
mkdir -p $OUT
export TMPDIR=/tmp
for v in A B; do
  crashes=0
  for i in 1 2 3 4 5 6; do
    if [ $v = A ]; then timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/run${v}_$i.log 2>&1; else DLWP_SPLITK=0 timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/run${v}_$i.log 2>&1; fi
    rc=$?
    echo "$v run $i rc=$rc $(grep -E 'passed|failed' $OUT/run${v}_$i.log | tail -1)" | tee -a $OUT/summary.txt
    if [ $rc -eq 139 ] || [ $rc -eq 134 ]; then crashes=$((crashes+1)); grep -n "Fatal\|File \"/.*repo" $OUT/run${v}_$i.log | head -6 >> $OUT/summary.txt; elif [ $rc -ne 0 ]; then grep -E "^FAILED" $OUT/run${v}_$i.log | head -5 >> $OUT/summary.txt; else rm -f $OUT/run${v}_$i.log; fi
  done
  echo "variant $v crashes: $crashes" | tee -a $OUT/summary.txt
done
cat $OUT/summary.txt
