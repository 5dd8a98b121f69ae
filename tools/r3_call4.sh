#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_fold.py tests/test_gpu_model.py tests/test_gpu_parallel.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log | head -2
for b in 8 16 32 64; do
  for g in 0 1; do
   DLWP_TRAIN_FOLD=1 DLWP_TRAIN_GRAPH=$g timeout 120 python tools/bench_train.py --batch $b --steps 40 --warmup 20 > $O/train_b${b}_graph${g}.json 2>$O/err.txt || tail -3 $O/err.txt
   python -c "
import json;d=json.loads(open('$O/train_b${b}_graph${g}.json').read().strip().splitlines()[-1]);print('b$b fold1 graph$g', round(d['ms_per_step'],4),'ms', round(d['value'],1))"
  done
done
bash tools/r3_trace.sh fg2 8 DLWP_TRAIN_FOLD=1 DLWP_TRAIN_GRAPH=1 > gpurun_out/r3t_fg2.txt 2>&1
