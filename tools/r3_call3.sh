#!/bin/bash
# round-3 GPU call 3: the folded training step -- parity, then timings (eager / graph, folded / unfolded, batch 8 / 64)
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_fold.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_parallel.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest.log
for b in 8 64; do
 for f in 1 0; do
  for g in 0 1; do
   DLWP_TRAIN_FOLD=$f DLWP_TRAIN_GRAPH=$g timeout 120 python tools/bench_train.py --batch $b --steps 40 --warmup 20 > $O/train_b${b}_fold${f}_graph${g}.json 2>$O/err.txt || tail -3 $O/err.txt
   python -c "
import json;d=json.loads(open('$O/train_b${b}_fold${f}_graph${g}.json').read().strip().splitlines()[-1]);print('b$b fold$f graph$g', round(d['ms_per_step'],4),'ms', round(d['value'],1))"
  done
 done
done
