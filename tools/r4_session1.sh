#!/bin/bash
# r4 GPU session 1: split-K parity + sweep + small-grid bench A/B.  usage (gpurun, repo root): bash tools/r4_session1.sh
OUT=gpurun_out/s1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_splitk.py -q -x > $OUT/tests_splitk.log 2>&1
echo "splitk tests rc=$?" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests_all.log 2>&1
echo "all gpu tests rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/tests_all.log >> $OUT/summary.txt
for m in 1 2 4 8 16; do
  extra=""; [ $m = 8 ] && extra="--all-configs"
  timeout 300 python tools/sweep_splitk.py --members $m $extra > $OUT/sweep_cfg2_m$m.json 2> $OUT/sweep_cfg2_m$m.err
done
timeout 400 python tools/sweep_splitk.py --grid 180x360 --channels 12 --members 4 --all-configs > $OUT/sweep_cfg5_m4.json 2> $OUT/sweep_cfg5_m4.err
for mode in 0 1; do
  for m in 1 8; do
    DLWP_SPLITK=$mode timeout 300 python bench.py --members $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_m${m}_splitk$mode.json 2>/dev/null
  done
  DLWP_SPLITK=$mode timeout 300 python bench.py --grid 180x360 --channels 12 --members 4 --forwards 40 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_cfg5_m4_splitk$mode.json 2>/dev/null
  DLWP_SPLITK=$mode timeout 300 python tools/bench_train.py --batch 8 --steps 50 --warmup 20 > $OUT/train_b8_splitk$mode.json 2>/dev/null
done
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
    else:
        print(f.split('/')[-1], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','ms_per_step','samples_per_s','unit','step_ms')})
PY
cat $OUT/summary.txt
