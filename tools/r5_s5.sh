#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parallel.py -m gpu -q -x -s > $O/par.log 2>&1; echo "parallel rc=$? $(grep -E 'passed|failed' $O/par.log | tail -1)"
grep -E "^FAILED|^ERROR|one-shot exchange region|Error" $O/par.log | head -20
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5e/bench.json').read().strip().splitlines()[-1])
s=d['sub_records']
print('value',round(d['value']), 'host_visible', {k:(round(v) if isinstance(v,float) and v>1000 else v) for k,v in s['host_visible'].items() if not isinstance(v,(dict,list,str))})
print('pad_hbm', [(r['kernel'],r['shape'][1:],r['hbm_frac']) for r in s['pad_hbm']['rows']])
print('cfg4', round(s['recurrent_cfg4_bf16']['value']), s['recurrent_cfg4_bf16'].get('null_stream'))
print('m1',round(s['members_1']['value']),'m8',round(s['members_8']['value']))
print('train', s['train_cfg3']['ms_per_step'], s['train_cfg3'].get('share_of_8_gpus',{}).get('ms_per_step'))
print('probes', d.get('stream_probes'))
print({k:v for k,v in s.items() if k.startswith('error')})
PY
