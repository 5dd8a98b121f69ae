#!/bin/bash
OUT=gpurun_out/s7
mkdir -p $OUT
export TMPDIR=/tmp
for b in 8 64; do
  timeout 300 python tools/bench_fit_generator.py --batch $b --samples $((b * 40)) --epochs 3 > $OUT/fitgen_b${b}_auto.json 2> $OUT/fitgen_b${b}_auto.err
  tail -1 $OUT/fitgen_b${b}_auto.json >> $OUT/summary.txt
  DLWP_TRAIN_STEP=graph timeout 300 python tools/bench_fit_generator.py --batch $b --samples $((b * 40)) --epochs 3 > $OUT/fitgen_b${b}_graph.json 2> $OUT/fitgen_b${b}_graph.err
  tail -1 $OUT/fitgen_b${b}_graph.json >> $OUT/summary.txt
done
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests_all.log 2>&1
echo "all gpu tests rc=$?" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR|passed|failed|Fatal" $OUT/tests_all.log | tail -30 >> $OUT/summary.txt
cp gpurun_out/forward_errors.json $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY >> $OUT/summary.txt
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
sr=d.get('sub_records',{})
for k,v in sr.items():
    print(k, json.dumps(v)[:1500])
PY
cat $OUT/summary.txt
