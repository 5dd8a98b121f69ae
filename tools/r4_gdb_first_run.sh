#!/bin/bash
OUT=gpurun_out/first/gdb
mkdir -p $OUT
timeout 1500 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "set confirm off" -ex "handle SIGUSR1 SIGUSR2 nostop noprint pass" -ex run -ex "bt 60" -ex "info sharedlibrary" --args python -m pytest tests -m gpu -q -p no:faulthandler -p no:cacheprovider > $OUT/gdb.log 2>&1
echo "gdb rc=$?" | tee $OUT/summary.txt
grep -n "SIGSEGV\|received signal" -A70 $OUT/gdb.log | head -120 | tee -a $OUT/summary.txt
tail -5 $OUT/gdb.log | cut -c1-200 >> $OUT/summary.txt
