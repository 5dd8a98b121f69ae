#!/bin/bash
# One-call measurement sweep for DESIGN.md section 6 (run through gpurun from the repo root):
#   batch sensitivity of the config-2 rollout, the config-5 share (1 deg, 12 channels), the config-3 training step.
# usage: bash tools/measure_round.sh r1e
TAG=${1:-meas}
OUT=gpurun_out/meas
mkdir -p $OUT
for m in 1 4 16 64; do
  python bench.py --members $m --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_m$m.json 2>/dev/null
done
python bench.py --grid 180x360 --channels 12 --members 4 --forwards 40 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_cfg5_1deg_4members.json 2>/dev/null
python bench.py --grid 180x360 --channels 12 --members 32 --forwards 40 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_cfg5_1deg_32members.json 2>/dev/null
python tools/bench_train.py --batch 64 --steps 20 --warmup 15 > $OUT/${TAG}_train_b64.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/${TAG}_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'ERR',e); continue
    print(f.split('/')[-1], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','ms_per_step','samples_per_s','unit')}, d.get('forward',{}).get('achieved_tflops'))
PY
