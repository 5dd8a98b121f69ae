#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16_octets.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_kernels.py -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 300 python tools/bench_cfg4.py > $O/cfg4.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('$O/cfg4.json').read().strip().splitlines()[-1])
print(round(d['six_hour_steps_per_s'],1), 'steps/s', round(d['ms_per_forward'],4), 'ms/fwd')
for r in d['launches']: print('   ', r['op'], r.get('layer',''), r.get('storage',''), r['ms'], r.get('frac_of_matrix_peak'), r.get('frac_of_hbm_peak'))
PY
timeout 600 python tools/tune_cfg4_octets.py > $O/tune.txt 2>$O/tune.err; grep "^{" $O/tune.txt | cut -c1-500 | head -3
