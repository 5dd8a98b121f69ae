#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_bf16_octets.py tests/test_gpu_kernels.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | head -1
echo base $(python tools/bench_layer1.py 2>/dev/null)
for k in 1 2 3 4; do echo knock$k $(DLWP_LIB_PATH=$PWD/dlwp_amd/knock/libdlwp_hip_f$k.so python tools/bench_layer1.py 2>/dev/null); done
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), [(l['layer'], l['ms']) for l in d['layers']])"
