#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | head -3
for b in 8 64; do
  timeout 120 python tools/bench_train.py --batch $b --steps 40 --warmup 20 > $O/train_b${b}.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/train_b${b}.json').read().strip().splitlines()[-1]);print('b$b', round(d['ms_per_step'],4),'ms', round(d['value'],1))"
done
