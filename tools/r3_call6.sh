#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $O/pytest.log
for o in 1 0; do
  DLWP_BF16_O8=$o timeout 300 python tools/bench_cfg4.py > $O/cfg4_o8_$o.json 2>$O/err$o.txt || tail -3 $O/err$o.txt
  python - <<PY
import json
d=json.loads(open('$O/cfg4_o8_$o.json').read().strip().splitlines()[-1])
print('O8=$o', round(d['six_hour_steps_per_s'],1), 'steps/s', round(d['ms_per_forward'],4), 'ms/forward', d['finite'])
PY
done
