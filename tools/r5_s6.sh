#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_fold.py tests/test_gpu_pair.py tests/test_gpu_configs.py -m gpu -q -x -k "gradient or train or pair or cfg3 or fold" > $O/t.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/t.log | tail -1)"
grep -E "^FAILED|^ERROR" $O/t.log | head
for b in 64 8; do for f in graph lanes; do DLWP_TRAIN_STEP=$f python tools/bench_train.py --batch $b --steps 40 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $b form $f', round(d['ms_per_step'],4), 'ms')"; done; done
cd /tmp
DLWP_TRAIN_STEP=graph rocprofv3 --kernel-trace --stats -d $O/tr -o s --output-format csv -- python $R/tools/bench_train.py --batch 64 --steps 40 --warmup 10 > /dev/null 2> $O/tr.err
grep -h "wgrad" $(find $O/tr -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4 | cut -c1-150
rm -rf $O/tr
DLWP_TRAIN_STEP=graph rocprofv3 --kernel-trace --stats -d $O/tr -o s --output-format csv -- python $R/tools/bench_train.py --batch 8 --steps 40 --warmup 10 > /dev/null 2> $O/tr.err
grep -h "wgrad" $(find $O/tr -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4 | cut -c1-150
rm -rf $O/tr
