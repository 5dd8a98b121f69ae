#!/bin/bash
# r4 GPU session 2: split-K (uncached slabs, no fences) parity + sweep over every split instance.
OUT=gpurun_out/s2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_splitk.py -q > $OUT/tests_splitk.log 2>&1
echo "splitk tests rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/tests_splitk.log >> $OUT/summary.txt
for m in 1 2 4 8 16; do
  timeout 400 python tools/sweep_splitk.py --members $m --all-configs --iters 100 > $OUT/sweep_cfg2_m$m.json 2> $OUT/sweep_cfg2_m$m.err
done
timeout 500 python tools/sweep_splitk.py --grid 180x360 --channels 12 --members 4 --all-configs --iters 100 > $OUT/sweep_cfg5_m4.json 2> $OUT/sweep_cfg5_m4.err
python - <<PY >> $OUT/summary.txt
import json,glob
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'ERR',e); continue
    if 'launches' in d:
        print(f.split('/')[-1], 'sum unsplit', d['sum_unsplit_us'], 'rule', d['sum_rule_us'], 'best', d['sum_best_us'])
        for r in d['launches']:
            print('   ', r['layer'], r['xs'], 'cout', r['cout'], 'cfg', r['cfg'], 'grid', r['grid'], 'unsplit', r['unsplit_us'], 'rule S', r['rule_S'], r['rule_us'], 'forced', r['forced_us'])
            for k, v in (r.get('other_configs_us') or {}).items():
                print('        other', k, v)
PY
cat $OUT/summary.txt
