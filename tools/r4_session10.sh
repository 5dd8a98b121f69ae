#!/bin/bash
OUT=gpurun_out/s10
mkdir -p $OUT
export TMPDIR=/tmp
crashes=0
for i in 1 2 3 4 5 6; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q > $OUT/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E 'passed|failed' $OUT/run_$i.log | tail -1)" | tee -a $OUT/summary.txt
  if [ $rc -eq 139 ] || [ $rc -eq 134 ]; then crashes=$((crashes+1)); grep -n "Fatal\|File \"/.*repo" $OUT/run_$i.log | head -6 >> $OUT/summary.txt; elif [ $rc -ne 0 ]; then grep -E "^FAILED|^E  " $OUT/run_$i.log | head -12 >> $OUT/summary.txt; else rm -f $OUT/run_$i.log; fi
done
echo "crashes: $crashes" | tee -a $OUT/summary.txt
cp gpurun_out/forward_errors.json $OUT/ 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY >> $OUT/summary.txt
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
sr=d.get('sub_records',{})
for k in ('members_1','members_8','layer1_at_91x180','train_cfg3','train_cfg3_loader_fed','error_local','error_collective'):
    if k in sr: print(k, json.dumps(sr[k])[:900])
PY
for b in 64 8; do timeout 300 python tools/bench_train.py --batch $b --steps 40 --warmup 20 2>/dev/null | tail -1 | cut -c1-300 >> $OUT/summary.txt; done
cat $OUT/summary.txt
