#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value'],1), 'roofline frac', round(d['roofline']['frac'],4), d['roofline'].get('rocprof'))
for k,v in d.get('sub_records',{}).items():
    if isinstance(v,dict): print('   ',k,{a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('value','frac','hbm_frac','ms_per_step','ms_per_forward','algorithmic_frac','bf16_mfma_frac')}, {a:{x:(round(y,3) if isinstance(y,float) else y) for x,y in b.items() if x in ('value','ms_per_step','frac') or x.startswith('projected')} for a,b in v.items() if isinstance(b,dict)})
    else: print('   ',k,v)
PY
