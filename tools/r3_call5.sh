#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16_octets.py -x -q > $O/pytest_oct.log 2>&1; echo "pytest oct rc=$?"
tail -30 $O/pytest_oct.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_kernels.py -q -k "bf16 or bfloat16 or cfg4 or lstm or recurrent" > $O/pytest_bf16.log 2>&1; echo "pytest bf16 rc=$?"
tail -8 $O/pytest_bf16.log
for o in 1 0; do
  DLWP_BF16_O8=$o timeout 300 python tools/bench_cfg4.py > $O/cfg4_o8_$o.json 2>$O/err$o.txt || tail -3 $O/err$o.txt
  python - <<PY
import json
d=json.loads(open('$O/cfg4_o8_$o.json').read().strip().splitlines()[-1])
print('O8=$o', round(d['six_hour_steps_per_s'],1), 'steps/s', round(d['ms_per_forward'],4), 'ms/forward', d['finite'])
for r in d['launches']: print('   ', r['op'], r.get('layer',''), r.get('storage',''), r['ms'], r.get('frac_of_matrix_peak'), r.get('frac_of_hbm_peak'))
PY
done
