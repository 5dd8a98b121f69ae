#!/bin/bash
# round-3 GPU call 1: parity + the new launcher tests, the bench line, small-batch baselines
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for b in 8 64; do
  timeout 120 python tools/bench_train.py --batch $b --steps 30 --warmup 20 > $O/train_b$b.json 2>/dev/null
  DLWP_TRAIN_GRAPH=1 timeout 120 python tools/bench_train.py --batch $b --steps 30 --warmup 20 > $O/train_graph_b$b.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'ERR',e); continue
    print(f.split('/')[-1], round(d['value'],1), d.get('ms_per_step'))
    for k,v in d.get('sub_records',{}).items():
        if isinstance(v,dict): print('   ',k,{a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('value','frac','hbm_frac','ms_per_step','ms_per_forward','algorithmic_frac','bf16_mfma_frac')}, {a:b.get('value') for a,b in v.items() if isinstance(b,dict) and 'value' in b}, {a:c for b in v.values() if isinstance(b,dict) for a,c in b.items() if a.startswith('projected')})
        else: print('   ',k,v)
PY
