#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16_octets.py tests/test_gpu_model.py tests/test_gpu_configs.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | head -2
show() { python - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', round(d['six_hour_steps_per_s'],1), 'steps/s', round(d['ms_per_forward'],4), 'ms/fwd', [ (r.get('layer','')[-10:], r['ms']) for r in d['launches'] if r.get('cell_update')])
PY
}
timeout 300 python tools/bench_cfg4.py > $O/cfg4_base.json 2>/dev/null; show $O/cfg4_base.json base
for k in 1 2 3 4 5; do
  DLWP_LIB_PATH=$PWD/dlwp_amd/knock/libdlwp_hip_k$k.so timeout 300 python tools/bench_cfg4.py > $O/cfg4_k$k.json 2>/dev/null; show $O/cfg4_k$k.json knock$k
done
timeout 600 python tools/tune_cfg4_octets.py > $O/tune.txt 2>$O/tune.err; grep "^{" $O/tune.txt | cut -c1-700
