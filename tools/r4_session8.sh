#!/bin/bash
OUT=gpurun_out/s8
mkdir -p $OUT
export TMPDIR=/tmp
GDB=/opt/rocm/bin/rocgdb
timeout 300 $GDB -batch -ex "set pagination off" -ex run -ex "bt 30" --args python -c "print('hello from python under gdb')" > $OUT/gdb_probe.log 2>&1
tail -5 $OUT/gdb_probe.log | tee $OUT/summary.txt
crashes=0
for i in $(seq 1 8); do
  timeout 600 python -X faulthandler -m pytest tests/test_gpu_model.py -q > $OUT/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E 'passed|failed' $OUT/run_$i.log | tail -1)" | tee -a $OUT/summary.txt
  if [ $rc -ne 0 ]; then crashes=$((crashes+1)); grep -n "Fatal\|File \"/" $OUT/run_$i.log | head -8 >> $OUT/summary.txt; else rm -f $OUT/run_$i.log; fi
done
echo "crashes with split-K on: $crashes" | tee -a $OUT/summary.txt
crashes=0
for i in $(seq 1 8); do
  DLWP_SPLITK=0 timeout 600 python -X faulthandler -m pytest tests/test_gpu_model.py -q -k "not batch_chunking" > $OUT/run0_$i.log 2>&1
  rc=$?
  echo "splitk0 run $i rc=$rc $(grep -E 'passed|failed' $OUT/run0_$i.log | tail -1)" | tee -a $OUT/summary.txt
  if [ $rc -ne 0 ]; then crashes=$((crashes+1)); grep -n "Fatal\|File \"/" $OUT/run0_$i.log | head -8 >> $OUT/summary.txt; else rm -f $OUT/run0_$i.log; fi
done
echo "crashes with split-K off: $crashes" | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_parallel.py -q > $OUT/parallel.log 2>&1
echo "parallel rc=$? $(grep -E 'passed|failed' $OUT/parallel.log | tail -1)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^E   " $OUT/parallel.log | head -20 >> $OUT/summary.txt
cat $OUT/summary.txt
